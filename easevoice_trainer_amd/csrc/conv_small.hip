// Degenerate Conv1d shapes of the s2 path that have no GEMM in them (gfx950):
//   * Cout == 1  (HiFi-GAN conv_post 16->1 k7, models.py:446; discriminator conv_post 1024->1 k3, models.py:536,574):
//     a dot product per output position  -> lane groups reduce over (tap, channel) with 16-byte loads;
//     its weight gradient is a dy-weighted sum of input rows -> per-thread register accumulators.
//   * Cin == 1   (discriminator first layers 1->16 k15 / 1->32 k5 s3, models.py:490-497,566): forward = a k-tap FIR per
//     output channel from an LDS-staged signal segment (8 channels = one 16-byte store per thread); backward-data = the
//     transposed FIR over an LDS tile of dy_eff; weight gradient = one thread per (channel, tap) over LDS tiles.
// HBM-bound byte work: coalesced 16-byte reads, LDS only for the small weight vector / block reduction.
#include "evt_common.h"
#include <cstdlib>
#include "../../include/evt.h"
#include "conv_p.h"

namespace {

struct SP {
  const void* x; const void* w; const float* bias; const void* y_in; const void* dy; void* y; float* dw;
  int nseq, lin, lout, cin, cout, k, stride, pad, dil;
  int ck, nchunk, kp;   // REG geometry
  float in_slope; int out_act; float out_slope;
  int G;                // lanes per output (power of two <= 64)
  int pos_per_block;
  float* ws;            // scratch rows for the per-block partial results (fold.hip), or null: fp32 atomics
  long ws_row;          // floats per scratch row
};

__device__ __forceinline__ long sreg_index(const SP& p, int d0, int d1, int t) {
  const int chunk = d1 / p.ck, cc = d1 - chunk * p.ck;
  return (((long)d0 * p.nchunk + chunk) * p.kp + t) * p.ck + cc;
}

// ---- Cout == 1 forward: G lanes per output --------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void cout1_fwd(SP p) {
  constexpr int V = 16 / sizeof(T);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* wl = reinterpret_cast<float*>(smem);   // [k][cin] fp32
  const T* w = reinterpret_cast<const T*>(p.w);
  for (int i = threadIdx.x; i < p.k * p.cin; i += 256) {
    const int t = i / p.cin, c = i - t * p.cin;
    wl[i] = to_f<T>(w[sreg_index(p, 0, c, t)]);
  }
  __syncthreads();
  const int G = p.G;
  const int sub = threadIdx.x % G;
  const long gid = ((long)blockIdx.x * 256 + threadIdx.x) / G;
  const long ngroups = (long)gridDim.x * 256 / G;
  const long total = (long)p.nseq * p.lout;
  const int ppr = p.cin / V;             // 16-byte pieces per row
  const int pieces = p.k * ppr;
  const T* x = reinterpret_cast<const T*>(p.x);
  const long rounds = (total + ngroups - 1) / ngroups;
  for (long rd = 0; rd < rounds; ++rd) {   // uniform trip count: the shuffles below need every lane
    const long o = rd * ngroups + gid;
    const bool live = o < total;
    const int q = live ? (int)(o % p.lout) : 0;
    const int seq = live ? (int)(o / p.lout) : 0;
    float acc = 0.f;
    if (live) {
      for (int pc = sub; pc < pieces; pc += G) {
        const int t = pc / ppr, c0 = (pc - t * ppr) * V;
        const int row = q * p.stride + t * p.dil - p.pad;
        if (row < 0 || row >= p.lin) continue;
        const uint4 v = *reinterpret_cast<const uint4*>(x + ((long)seq * p.lin + row) * p.cin + c0);
        const T* pv = reinterpret_cast<const T*>(&v);
        const float* wr = wl + t * p.cin + c0;
#pragma unroll
        for (int e = 0; e < V; ++e) acc += lrelu_f(to_f<T>(pv[e]), p.in_slope) * wr[e];
      }
    }
    for (int off = G >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (live && sub == 0) {
      if (p.bias) acc += p.bias[0];
      if (p.out_act == EVT_ACT_LRELU) acc = lrelu_f(acc, p.out_slope);
      else if (p.out_act == EVT_ACT_TANH) acc = tanhf(acc);
      reinterpret_cast<T*>(p.y)[o] = from_f<T>(acc);
    }
  }
}

// ---- Cout == 1 backward-weight: dW[t][c] += sum_pos dy_eff[pos] * lrelu(x)[row(pos,t)][c] --------------------
// thread -> (position lane pl, piece pc); up to NA pieces per thread; block covers pos_per_block positions
template <typename T>
__global__ __launch_bounds__(256) void cout1_bwd_weight(SP p) {
  constexpr int V = 16 / sizeof(T);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* red = reinterpret_cast<float*>(smem);   // [npl][k*cin]
  const int ppr = p.cin / V;
  const int pieces = p.k * ppr;
  int pp = 1;
  while (pp < pieces && pp < 256) pp <<= 1;       // pieces padded to a power of two (<= 256)
  const int npl = 256 / pp;                        // positions processed in parallel
  const int pl = threadIdx.x / pp, pc0 = threadIdx.x % pp;
  constexpr int NA = 4;   // k*cin <= 4096 (evt_small_kind) -> at most 1024 pieces -> 4 per thread
  float acc[NA][V];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int e = 0; e < V; ++e) acc[a][e] = 0.f;
  const T* x = reinterpret_cast<const T*>(p.x);
  const T* dy = reinterpret_cast<const T*>(p.dy);
  const T* ys = reinterpret_cast<const T*>(p.y_in);
  const long total = (long)p.nseq * p.lout;
  const long p0 = (long)blockIdx.x * p.pos_per_block;
  const long p1 = min(total, p0 + p.pos_per_block);
  // UP positions per trip: all their 16-byte loads are issued before the first use (the loop is latency-bound)
  constexpr int UP = 4;
  for (long ob = p0 + pl; ob < p1; ob += (long)npl * UP) {
    uint4 v[UP][NA];
    float d[UP];
#pragma unroll
    for (int u = 0; u < UP; ++u) {
      const long o = ob + (long)u * npl;
      const bool live = o < p1;
      const int q = live ? (int)(o % p.lout) : 0, seq = live ? (int)(o / p.lout) : 0;
      d[u] = live ? to_f<T>(dy[o]) : 0.f;
      if (live && ys) d[u] *= dact_from_out(p.out_act, to_f<T>(ys[o]), p.out_slope);
#pragma unroll
      for (int a = 0; a < NA; ++a) {
        const int pc = pc0 + a * pp;
        const int t = pc / ppr, c0 = (pc - t * ppr) * V;
        const int row = q * p.stride + t * p.dil - p.pad;
        const bool ok = live && pc < pieces && row >= 0 && row < p.lin;
        v[u][a] = ok ? *reinterpret_cast<const uint4*>(x + ((long)seq * p.lin + row) * p.cin + c0) : make_uint4(0, 0, 0, 0);
      }
    }
#pragma unroll
    for (int u = 0; u < UP; ++u)
#pragma unroll
      for (int a = 0; a < NA; ++a) {
        const T* pv = reinterpret_cast<const T*>(&v[u][a]);
#pragma unroll
        for (int e = 0; e < V; ++e) acc[a][e] += d[u] * lrelu_f(to_f<T>(pv[e]), p.in_slope);
      }
  }
  // the npl position lanes of the block meet in LDS and are added in lane order (LDS atomics would add them in arrival
  // order: last bits that change from run to run)
  const int kc = p.k * p.cin;
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    const int pc = pc0 + a * pp;
    if (pc < pieces) {
      const int t = pc / ppr, c0 = (pc - t * ppr) * V;
#pragma unroll
      for (int e = 0; e < V; ++e) red[pl * kc + t * p.cin + c0 + e] = acc[a][e];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kc; i += 256) {
    const int t = i / p.cin, c = i - t * p.cin;
    float v = red[i];
    for (int l = 1; l < npl; ++l) v += red[l * kc + i];
    if (p.ws) p.ws[(long)blockIdx.x * p.ws_row + sreg_index(p, 0, c, t)] = v;
    else atomicAdd(p.dw + sreg_index(p, 0, c, t), v);
  }
}

// ---- Cin == 1 forward: y[q][co] = act(b[co] + sum_t w[co][t] * lrelu(x)[q*s + t*dil - pad]) ---------------------
// block = TP consecutive outputs of one sequence; thread = 8 channels of one position per pass
constexpr int C1_TP = 512;

template <typename T>
__global__ __launch_bounds__(256) void cin1_fwd(SP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* wl = reinterpret_cast<float*>(smem);            // [k][cout]
  float* xl = wl + p.k * p.cout;                          // signal segment
  const T* w = reinterpret_cast<const T*>(p.w);
  for (int i = threadIdx.x; i < p.k * p.cout; i += 256) {
    const int t = i / p.cout, co = i - t * p.cout;
    wl[i] = to_f<T>(w[sreg_index(p, co, 0, t)]);
  }
  const int tiles = (p.lout + C1_TP - 1) / C1_TP;
  const int seq = blockIdx.x / tiles, q0 = (blockIdx.x - seq * tiles) * C1_TP;
  const int nq = min(C1_TP, p.lout - q0);
  const int seg = (nq - 1) * p.stride + (p.k - 1) * p.dil + 1;
  const int r0 = q0 * p.stride - p.pad;
  const T* x = reinterpret_cast<const T*>(p.x) + (long)seq * p.lin;
  for (int i = threadIdx.x; i < seg; i += 256) {
    const int r = r0 + i;
    xl[i] = (r >= 0 && r < p.lin) ? lrelu_f(to_f<T>(x[r]), p.in_slope) : 0.f;
  }
  __syncthreads();
  const int CG = p.cout >> 3, ppp = 256 / CG;
  const int c0 = (threadIdx.x % CG) * 8;
  float bv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bv[e] = p.bias ? p.bias[c0 + e] : 0.f;
  T* y = reinterpret_cast<T*>(p.y) + ((long)seq * p.lout + q0) * p.cout;
  for (int q = threadIdx.x / CG; q < nq; q += ppp) {
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = bv[e];
    for (int t = 0; t < p.k; ++t) {
      const float xv = xl[q * p.stride + t * p.dil];
      const float4 w0 = *reinterpret_cast<const float4*>(wl + t * p.cout + c0);
      const float4 w1 = *reinterpret_cast<const float4*>(wl + t * p.cout + c0 + 4);
      acc[0] += xv * w0.x; acc[1] += xv * w0.y; acc[2] += xv * w0.z; acc[3] += xv * w0.w;
      acc[4] += xv * w1.x; acc[5] += xv * w1.y; acc[6] += xv * w1.z; acc[7] += xv * w1.w;
    }
    T outv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = acc[e];
      if (p.out_act == EVT_ACT_LRELU) v = lrelu_f(v, p.out_slope);
      else if (p.out_act == EVT_ACT_TANH) v = tanhf(v);
      outv[e] = from_f<T>(v);
    }
    T* dst = y + (long)q * p.cout + c0;
    if constexpr (sizeof(T) == 2) *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<uint4*>(outv);
    else { reinterpret_cast<float4*>(dst)[0] = reinterpret_cast<float4*>(outv)[0]; reinterpret_cast<float4*>(dst)[1] = reinterpret_cast<float4*>(outv)[1]; }
  }
}

// ---- Cin == 1 backward-data: dx[i] = sum_t sum_co dy_eff[(i + pad - t*dil)/s][co] * w[co][t] (exact multiples only) --
constexpr int C1_TI = 512;

template <typename T>
__global__ __launch_bounds__(256) void cin1_bwd_data(SP p, const void* gate, const void* dx_add) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* wl = reinterpret_cast<float*>(smem);            // [k][cout]
  float* dyl = wl + p.k * p.cout;                         // [rows][cout + 1]
  const int pitch = p.cout + 1;
  const T* w = reinterpret_cast<const T*>(p.w);
  for (int i = threadIdx.x; i < p.k * p.cout; i += 256) {
    const int t = i / p.cout, co = i - t * p.cout;
    wl[i] = to_f<T>(w[sreg_index(p, co, 0, t)]);
  }
  const int tiles = (p.lin + C1_TI - 1) / C1_TI;
  const int seq = blockIdx.x / tiles, i0 = (blockIdx.x - seq * tiles) * C1_TI;
  const int ni = min(C1_TI, p.lin - i0);
  // output rows q that can touch inputs [i0, i0 + ni): q*s - pad + t*dil = i
  int qlo = i0 + p.pad - (p.k - 1) * p.dil;
  qlo = qlo <= 0 ? 0 : (qlo + p.stride - 1) / p.stride;
  int qhi = (i0 + ni - 1 + p.pad) / p.stride;
  if (qhi > p.lout - 1) qhi = p.lout - 1;
  const int nrows = qhi - qlo + 1;
  const T* dy = reinterpret_cast<const T*>(p.dy) + ((long)seq * p.lout + qlo) * p.cout;
  const T* ys = p.y_in ? reinterpret_cast<const T*>(p.y_in) + ((long)seq * p.lout + qlo) * p.cout : nullptr;
  constexpr int V = 16 / sizeof(T);
  for (int i = threadIdx.x; i < nrows * p.cout / V; i += 256) {   // cout % 8 == 0: pieces never straddle rows
    const uint4 v = reinterpret_cast<const uint4*>(dy)[i];
    uint4 va = make_uint4(0, 0, 0, 0);
    if (ys) va = reinterpret_cast<const uint4*>(ys)[i];
    const T* pv = reinterpret_cast<const T*>(&v);
    const T* pa = reinterpret_cast<const T*>(&va);
    const int r = (i * V) / p.cout, c = i * V - r * p.cout;
#pragma unroll
    for (int e = 0; e < V; ++e) {
      float d = to_f<T>(pv[e]);
      if (ys) d *= dact_from_out(p.out_act, to_f<T>(pa[e]), p.out_slope);
      dyl[r * pitch + c + e] = d;
    }
  }
  __syncthreads();
  T* dx = reinterpret_cast<T*>(p.y) + (long)seq * p.lin;
  const T* gt = gate ? reinterpret_cast<const T*>(gate) + (long)seq * p.lin : nullptr;
  const T* ad = dx_add ? reinterpret_cast<const T*>(dx_add) + (long)seq * p.lin : nullptr;
  for (int ii = threadIdx.x; ii < ni; ii += 256) {
    const int i = i0 + ii;
    float acc = 0.f;
    for (int t = 0; t < p.k; ++t) {
      const int j = i + p.pad - t * p.dil;
      if (j < 0) break;
      const int q = j / p.stride;
      if (q * p.stride != j || q > qhi || q < qlo) continue;
      const float* dr = dyl + (q - qlo) * pitch;
      const float* wr = wl + t * p.cout;
      for (int c = 0; c < p.cout; ++c) acc += dr[c] * wr[c];
    }
    if (gt) acc *= (to_f<T>(gt[i]) > 0.f ? 1.f : p.in_slope);
    if (ad) acc += to_f<T>(ad[i]);
    dx[i] = from_f<T>(acc);
  }
}

// ---- Cin == 1 backward-weight: dW[co][t] += sum_pos dy_eff[pos][co] * lrelu(x)[pos*s + t*dil - pad]; dbias fused ----
// one thread per (co, t) (+ cout threads for dbias) over LDS tiles of C1_TQ positions; blocks stride over the tiles
constexpr int C1_TQ = 256;

template <typename T>
__global__ __launch_bounds__(256) void cin1_bwd_weight(SP p, float* dbias) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* dyl = reinterpret_cast<float*>(smem);           // [C1_TQ][cout]
  float* xl = dyl + C1_TQ * p.cout;                       // signal segment
  const int nkt = p.cout * p.k;
  const int role = threadIdx.x < nkt ? 0 : (threadIdx.x < nkt + p.cout ? 1 : 2);
  const int co = role == 0 ? threadIdx.x % p.cout : (role == 1 ? threadIdx.x - nkt : 0);
  const int t = role == 0 ? threadIdx.x / p.cout : 0;
  const int tiles = (p.lout + C1_TQ - 1) / C1_TQ;
  const long ntiles = (long)p.nseq * tiles;
  float acc = 0.f;
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int seq = (int)(tile / tiles), q0 = (int)(tile - (long)seq * tiles) * C1_TQ;
    const int nq = min(C1_TQ, p.lout - q0);
    const int seg = (nq - 1) * p.stride + (p.k - 1) * p.dil + 1;
    const int r0 = q0 * p.stride - p.pad;
    const T* x = reinterpret_cast<const T*>(p.x) + (long)seq * p.lin;
    const T* dy = reinterpret_cast<const T*>(p.dy) + ((long)seq * p.lout + q0) * p.cout;
    const T* ys = p.y_in ? reinterpret_cast<const T*>(p.y_in) + ((long)seq * p.lout + q0) * p.cout : nullptr;
    __syncthreads();
    for (int i = threadIdx.x; i < seg; i += 256) {
      const int r = r0 + i;
      xl[i] = (r >= 0 && r < p.lin) ? lrelu_f(to_f<T>(x[r]), p.in_slope) : 0.f;
    }
    constexpr int V = 16 / sizeof(T);
    for (int i = threadIdx.x; i < nq * p.cout / V; i += 256) {   // cout % 8 == 0: 16-byte pieces never straddle rows
      const uint4 v = reinterpret_cast<const uint4*>(dy)[i];
      uint4 va = make_uint4(0, 0, 0, 0);
      if (ys) va = reinterpret_cast<const uint4*>(ys)[i];
      const T* pv = reinterpret_cast<const T*>(&v);
      const T* pa = reinterpret_cast<const T*>(&va);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        float d = to_f<T>(pv[e]);
        if (ys) d *= dact_from_out(p.out_act, to_f<T>(pa[e]), p.out_slope);
        dyl[i * V + e] = d;
      }
    }
    __syncthreads();
    if (role == 0) {
      // four independent chains, 8 positions per trip: the loop is LDS-latency-bound, not FMA-bound
      const float* xr = xl + t * p.dil;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      int q = 0;
      for (; q + 8 <= nq; q += 8) {
        float dv[8], xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { dv[u] = dyl[(q + u) * p.cout + co]; xv[u] = xr[(q + u) * p.stride]; }
        a0 += dv[0] * xv[0] + dv[4] * xv[4];
        a1 += dv[1] * xv[1] + dv[5] * xv[5];
        a2 += dv[2] * xv[2] + dv[6] * xv[6];
        a3 += dv[3] * xv[3] + dv[7] * xv[7];
      }
      for (; q < nq; ++q) a0 += dyl[q * p.cout + co] * xr[q * p.stride];
      acc += (a0 + a1) + (a2 + a3);
    } else if (role == 1) {
      float a0 = 0.f, a1 = 0.f;
      int q = 0;
      for (; q + 2 <= nq; q += 2) { a0 += dyl[q * p.cout + co]; a1 += dyl[(q + 1) * p.cout + co]; }
      if (q < nq) a0 += dyl[q * p.cout + co];
      acc += a0 + a1;
    }
  }
  if (p.ws) {
    // scratch row = [the dW image | cout bias sums]
    if (role == 0) p.ws[(long)blockIdx.x * p.ws_row + sreg_index(p, co, 0, t)] = acc;
    else if (role == 1) p.ws[(long)blockIdx.x * p.ws_row + (p.ws_row - p.cout) + co] = acc;
  } else {
    if (role == 0) atomicAdd(p.dw + sreg_index(p, co, 0, t), acc);
    else if (role == 1 && dbias) atomicAdd(dbias + co, acc);
  }
}

// ---- Cout == 1 backward-data: dx[i][c] = sum_t dy_eff[(i + pad - t*dil)/s] * w[t][c]; thread = 16 bytes of dx ---------
template <typename T>
__global__ __launch_bounds__(256) void cout1_bwd_data(SP p, const void* gate, const void* dx_add) {
  constexpr int V = 16 / sizeof(T);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* wl = reinterpret_cast<float*>(smem);   // [k][cin]
  const T* w = reinterpret_cast<const T*>(p.w);
  for (int i = threadIdx.x; i < p.k * p.cin; i += 256) {
    const int t = i / p.cin, c = i - t * p.cin;
    wl[i] = to_f<T>(w[sreg_index(p, 0, c, t)]);
  }
  __syncthreads();
  const int ppr = p.cin / V;
  const long total = (long)p.nseq * p.lin * ppr;
  const T* dy = reinterpret_cast<const T*>(p.dy);
  const T* ys = reinterpret_cast<const T*>(p.y_in);
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long row = idx / ppr;
    const int c0 = (int)(idx - row * ppr) * V;
    const int seq = (int)(row / p.lin), i = (int)(row - (long)seq * p.lin);
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
    for (int t = 0; t < p.k; ++t) {
      const int j = i + p.pad - t * p.dil;
      if (j < 0) break;
      const int q = j / p.stride;
      if (q * p.stride != j || q >= p.lout) continue;
      const long o = (long)seq * p.lout + q;
      float d = to_f<T>(dy[o]);
      if (ys) d *= dact_from_out(p.out_act, to_f<T>(ys[o]), p.out_slope);
      const float* wr = wl + t * p.cin + c0;
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] += d * wr[e];
    }
    const long off = row * p.cin + c0;
    T outv[V];
    uint4 gv = make_uint4(0, 0, 0, 0), av = make_uint4(0, 0, 0, 0);
    if (gate) gv = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(gate) + off);
    if (dx_add) av = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(dx_add) + off);
    const T* pg = reinterpret_cast<const T*>(&gv);
    const T* pa = reinterpret_cast<const T*>(&av);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      float v = acc[e];
      if (gate) v *= (to_f<T>(pg[e]) > 0.f ? 1.f : p.in_slope);
      if (dx_add) v += to_f<T>(pa[e]);
      outv[e] = from_f<T>(v);
    }
    *reinterpret_cast<uint4*>(reinterpret_cast<T*>(p.y) + off) = *reinterpret_cast<uint4*>(outv);
  }
}

SP make_sp(const evt_conv1d_params* c) {
  SP p{};
  p.nseq = c->nseq; p.lin = c->lin; p.lout = evt_conv1d_lout(c); p.cin = c->cin; p.cout = c->cout; p.k = c->k;
  p.stride = c->stride; p.pad = c->pad; p.dil = c->dil; p.in_slope = c->in_slope; p.out_act = c->out_act;
  p.out_slope = c->out_slope;
  evt_wlayout l; evt_conv1d_layout(c, &l);
  p.ck = l.reg_ck; p.nchunk = l.reg_nchunk; p.kp = l.reg_kp;
  return p;
}

// ---- Cout == 1 weight gradient, x-stationary form (stride 1, dilation 1): a thread keeps ONE 16-byte piece of an input
//      row and adds it into the K taps it belongs to (output positions q = i + pad - t), so every input row is read once
//      instead of once per tap, and the per-tap dy values are 4-byte loads that hit L1.  K * V accumulators per thread.
template <typename T, int K>
__global__ __launch_bounds__(256) void cout1_bwd_weight_xs(SP p, int rows_per_block) {
  constexpr int V = 16 / sizeof(T);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* red = reinterpret_cast<float*>(smem);   // [npl][K * cin]
  const int ppr = p.cin / V;
  int pp = 1;
  while (pp < ppr) pp <<= 1;                       // pieces per row padded to a power of two (<= 256)
  const int npl = 256 / pp;                        // rows in flight per block trip
  const int rl = threadIdx.x / pp, pc = threadIdx.x % pp;
  float acc[K][V];
#pragma unroll
  for (int t = 0; t < K; ++t)
#pragma unroll
    for (int e = 0; e < V; ++e) acc[t][e] = 0.f;
  const T* x = reinterpret_cast<const T*>(p.x);
  const T* dy = reinterpret_cast<const T*>(p.dy);
  const T* ys = reinterpret_cast<const T*>(p.y_in);
  const long total = (long)p.nseq * p.lin;
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = min(total, r0 + rows_per_block);
  constexpr int UP = 4;                            // rows per thread and trip: their loads are issued before the first use
  if (pc < ppr) {
    for (long rb = r0 + rl; rb < r1; rb += (long)npl * UP) {
      uint4 v[UP];
      float d[UP][K];
#pragma unroll
      for (int u = 0; u < UP; ++u) {
        const long r = rb + (long)u * npl;
        const bool live = r < r1;
        const long seq = live ? r / p.lin : 0;
        const int i = live ? (int)(r - seq * p.lin) : 0;
        v[u] = live ? *reinterpret_cast<const uint4*>(x + r * p.cin + pc * V) : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < K; ++t) {
          const int q = i + p.pad - t;
          const bool ok = live && q >= 0 && q < p.lout;
          const long o = seq * p.lout + (ok ? q : 0);
          float dv = ok ? to_f<T>(dy[o]) : 0.f;
          if (ok && ys) dv *= dact_from_out(p.out_act, to_f<T>(ys[o]), p.out_slope);
          d[u][t] = dv;
        }
      }
#pragma unroll
      for (int u = 0; u < UP; ++u) {
        const T* pv = reinterpret_cast<const T*>(&v[u]);
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const float xv = lrelu_f(to_f<T>(pv[e]), p.in_slope);
#pragma unroll
          for (int t = 0; t < K; ++t) acc[t][e] += d[u][t] * xv;
        }
      }
    }
  }
  // the row lanes of the block meet in LDS and are added in lane order (deterministic)
  const int kc = K * p.cin;
  if (pc < ppr) {
#pragma unroll
    for (int t = 0; t < K; ++t)
#pragma unroll
      for (int e = 0; e < V; ++e) red[rl * kc + t * p.cin + pc * V + e] = acc[t][e];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kc; i += 256) {
    const int t = i / p.cin, c = i - t * p.cin;
    float v = red[i];
    for (int l = 1; l < npl; ++l) v += red[l * kc + i];
    if (p.ws) p.ws[(long)blockIdx.x * p.ws_row + sreg_index(p, 0, c, t)] = v;
    else atomicAdd(p.dw + sreg_index(p, 0, c, t), v);
  }
}

}  // namespace

extern "C" int evt_small_kind(const evt_conv1d_params* c) {
  if (c->transposed || c->groups != 1) return 0;
  const int V = c->dtype == EVT_DT_HALF ? 8 : 4;
  if (c->cout == 1 && c->cin % V == 0 && (long)c->k * c->cin <= 4096) return 1;   // dot-product conv
  if (c->cin == 1 && c->cout % 8 == 0 && c->cout <= 32 && 256 % c->cout == 0 && c->cout * (c->k + 1) <= 256)
    return 2;  // single-channel input
  return 0;
}

extern "C" int evt_cout1_fwd(const evt_conv1d_params* c, const void* x, const void* w_reg, const float* bias, void* y,
                             void* stream) {
  SP p = make_sp(c);
  p.x = x; p.w = w_reg; p.bias = bias; p.y = y;
  const int V = c->dtype == EVT_DT_HALF ? 8 : 4;
  const int pieces = c->k * (c->cin / V);
  int G = 1;
  while (G < pieces && G < 64) G <<= 1;
  p.G = G;
  const long total = (long)p.nseq * p.lout;
  long blocks = (total * G + 255) / 256;
  static const long cap = getenv("EVT_COUT1_FWD_CAP") ? atol(getenv("EVT_COUT1_FWD_CAP")) : 4096;   // measurement knob
  if (blocks > cap) blocks = cap;
  const size_t lds = (size_t)c->k * c->cin * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  evt_set_last_tag("cout1_fwd");
  if (c->dtype == EVT_DT_HALF) hipLaunchKernelGGL(cout1_fwd<h16_t>, dim3((int)blocks), dim3(256), lds, st, p);
  else hipLaunchKernelGGL(cout1_fwd<float>, dim3((int)blocks), dim3(256), lds, st, p);
  return evt_check_launch();
}

extern "C" int evt_cout1_bwd_weight(const evt_conv1d_params* c, const void* x, const void* dy, const void* y, float* dw,
                                    float* ws, long ws_floats, void* stream) {
  SP p = make_sp(c);
  p.x = x; p.dy = dy; p.y_in = c->out_act != EVT_ACT_NONE ? y : nullptr; p.dw = dw;
  const long total = (long)p.nseq * p.lout;
  long ppb = (total + 255) / 256;   // ~256 blocks: enough loads in flight, a bounded number of partial results
  if (ppb < 16) ppb = 16;
  const long img = (long)p.nchunk * p.kp * p.ck;          // d0 = 1: one row of the image
  if (ws && img * 2 <= ws_floats) {
    // partial rows instead of atomics: the block count is no longer bounded by same-address atomics, and the loop is
    // latency-bound (a trip = 4 positions per lane) -- four times the blocks, a quarter of the trips
    static const long tgt = getenv("EVT_COUT1_WG_BLOCKS") ? atol(getenv("EVT_COUT1_WG_BLOCKS")) : 1024;   // measurement knob
    static const long minp = getenv("EVT_COUT1_WG_MINP") ? atol(getenv("EVT_COUT1_WG_MINP")) : 16;
    ppb = (total + tgt - 1) / tgt;
    if (ppb < minp) ppb = minp;
    const long maxb = ws_floats / img;
    if ((total + ppb - 1) / ppb > maxb) ppb = (total + maxb - 1) / maxb;
    p.ws = ws; p.ws_row = img;
  }
  p.pos_per_block = (int)ppb;
  int blocks = (int)((total + ppb - 1) / ppb);
  const int V = c->dtype == EVT_DT_HALF ? 8 : 4;
  hipStream_t st = (hipStream_t)stream;
  static const bool xs_off = getenv("EVT_NO_COUT1_XS") != nullptr;          // A/B switch
  int ppr2 = 1;
  while (ppr2 < c->cin / V) ppr2 <<= 1;
  const size_t lds_xs = (size_t)(256 / ppr2) * c->k * c->cin * sizeof(float);
  if (!xs_off && p.stride == 1 && p.dil == 1 && (c->k == 3 || c->k == 7) && ppr2 <= 256 && lds_xs <= (60u << 10)) {
    // x-stationary form: blocks over INPUT rows (same bounds on the block count as below)
    const long rows = (long)p.nseq * p.lin;
    long rpb = (rows + blocks - 1) / blocks;
    if (rpb < 16) rpb = 16;
    blocks = (int)((rows + rpb - 1) / rpb);
    evt_set_last_tag("cout1_bwd_weight_xs<k%d>", c->k);
#define XS(T, K_) hipLaunchKernelGGL((cout1_bwd_weight_xs<T, K_>), dim3(blocks), dim3(256), lds_xs, st, p, (int)rpb)
    if (c->dtype == EVT_DT_HALF) { if (c->k == 3) XS(h16_t, 3); else XS(h16_t, 7); }
    else { if (c->k == 3) XS(float, 3); else XS(float, 7); }
#undef XS
  } else {
    int pp = 1;
    while (pp < c->k * (c->cin / V) && pp < 256) pp <<= 1;
    const size_t lds = (size_t)(256 / pp) * c->k * c->cin * sizeof(float);    // one partial per position lane
    evt_set_last_tag("cout1_bwd_weight");
    if (c->dtype == EVT_DT_HALF) hipLaunchKernelGGL(cout1_bwd_weight<h16_t>, dim3(blocks), dim3(256), lds, st, p);
    else hipLaunchKernelGGL(cout1_bwd_weight<float>, dim3(blocks), dim3(256), lds, st, p);
  }
  int rc = evt_check_launch();
  if (rc || !p.ws) return rc;
  // the image's padded entries (kp > k, ck > cin) are never written by the blocks: only the k * cin live ones are folded
  // -- they are contiguous per (chunk, tap) run of ck floats, and for the shapes evt_small_kind admits (cin % 8 == 0) the
  // image has no padding at all (ck divides cin, kp == k for ck == 32, even-padded for ck == 16)
  if (img != (long)c->k * c->cin) {
    // padded taps: fold run by run
    for (int chk = 0; chk < p.nchunk && !rc; ++chk)
      rc = evt_conv::launch_fold_partials(p.ws + (long)chk * p.kp * p.ck, img, blocks, dw + (long)chk * p.kp * p.ck,
                                          (long)c->k * p.ck, st);
    return rc;
  }
  return evt_conv::launch_fold_partials(p.ws, img, blocks, dw, img, st);
}

extern "C" int evt_cin1_bwd_weight(const evt_conv1d_params* c, const void* x, const void* dy, const void* y, float* dw,
                                   float* dbias, float* ws, long ws_floats, void* stream) {
  SP p = make_sp(c);
  p.x = x; p.dy = dy; p.y_in = c->out_act != EVT_ACT_NONE ? y : nullptr; p.dw = dw;
  const long ntiles = (long)p.nseq * ((p.lout + C1_TQ - 1) / C1_TQ);
  int blocks = (int)(ntiles < 512 ? ntiles : 512);   // <= 512 partial results per dW element
  // cin = 1: the image is [cout][1][kp][1]; a scratch row holds it and the bias sums
  const long img = (long)p.cout * p.nchunk * p.kp * p.ck;
  const long row = img + p.cout;
  if (ws && blocks >= 2 && row * 2 <= ws_floats) {
    if (row * blocks > ws_floats) blocks = (int)(ws_floats / row);
    p.ws = ws; p.ws_row = row;
  }
  const int seg = (C1_TQ - 1) * c->stride + (c->k - 1) * c->dil + 1;
  const size_t lds = ((size_t)C1_TQ * c->cout + seg) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  evt_set_last_tag("cin1_bwd_weight");
  if (c->dtype == EVT_DT_HALF) hipLaunchKernelGGL(cin1_bwd_weight<h16_t>, dim3(blocks), dim3(256), lds, st, p, dbias);
  else hipLaunchKernelGGL(cin1_bwd_weight<float>, dim3(blocks), dim3(256), lds, st, p, dbias);
  int rc = evt_check_launch();
  if (rc || !p.ws) return rc;
  // live entries: taps t < k of every output channel (kp may be padded): fold the whole image when it has no padding
  if (p.kp == c->k) rc = evt_conv::launch_fold_partials(p.ws, row, blocks, dw, img, st);
  else
    for (int co = 0; co < p.cout && !rc; ++co)
      rc = evt_conv::launch_fold_partials(p.ws + (long)co * p.kp, row, blocks, dw + (long)co * p.kp, c->k, st);
  if (rc || !dbias) return rc;
  return evt_conv::launch_fold_partials(p.ws + img, row, blocks, dbias, p.cout, st);
}

extern "C" int evt_cin1_fwd(const evt_conv1d_params* c, const void* x, const void* w_reg, const float* bias, void* y,
                            void* stream) {
  SP p = make_sp(c);
  p.x = x; p.w = w_reg; p.bias = bias; p.y = y;
  const int blocks = p.nseq * ((p.lout + C1_TP - 1) / C1_TP);
  const int seg = (C1_TP - 1) * c->stride + (c->k - 1) * c->dil + 1;
  const size_t lds = ((size_t)c->k * c->cout + seg) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  evt_set_last_tag("cin1_fwd");
  if (c->dtype == EVT_DT_HALF) hipLaunchKernelGGL(cin1_fwd<h16_t>, dim3(blocks), dim3(256), lds, st, p);
  else hipLaunchKernelGGL(cin1_fwd<float>, dim3(blocks), dim3(256), lds, st, p);
  return evt_check_launch();
}

extern "C" int evt_cin1_bwd_data(const evt_conv1d_params* c, const void* dy, const void* y, const void* w_reg,
                                 const void* gate, const void* dx_add, void* dx, void* stream) {
  SP p = make_sp(c);
  p.dy = dy; p.y_in = c->out_act != EVT_ACT_NONE ? y : nullptr; p.w = w_reg; p.y = dx;
  const int blocks = p.nseq * ((p.lin + C1_TI - 1) / C1_TI);
  const int rows = (C1_TI + (c->k - 1) * c->dil) / c->stride + 2;
  const size_t lds = ((size_t)c->k * c->cout + (size_t)rows * (c->cout + 1)) * sizeof(float);
  if (lds > 64 * 1024) return EVT_ENOTSUP;
  hipStream_t st = (hipStream_t)stream;
  evt_set_last_tag("cin1_bwd_data");
  if (c->dtype == EVT_DT_HALF) hipLaunchKernelGGL(cin1_bwd_data<h16_t>, dim3(blocks), dim3(256), lds, st, p, gate, dx_add);
  else hipLaunchKernelGGL(cin1_bwd_data<float>, dim3(blocks), dim3(256), lds, st, p, gate, dx_add);
  return evt_check_launch();
}

extern "C" int evt_cout1_bwd_data(const evt_conv1d_params* c, const void* dy, const void* y, const void* w_reg,
                                  const void* gate, const void* dx_add, void* dx, void* stream) {
  SP p = make_sp(c);
  p.dy = dy; p.y_in = c->out_act != EVT_ACT_NONE ? y : nullptr; p.w = w_reg; p.y = dx;
  const int V = c->dtype == EVT_DT_HALF ? 8 : 4;
  const long total = (long)p.nseq * p.lin * (p.cin / V);
  long blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  const size_t lds = (size_t)c->k * c->cin * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  evt_set_last_tag("cout1_bwd_data");
  if (c->dtype == EVT_DT_HALF) hipLaunchKernelGGL(cout1_bwd_data<h16_t>, dim3((int)blocks), dim3(256), lds, st, p, gate, dx_add);
  else hipLaunchKernelGGL(cout1_bwd_data<float>, dim3((int)blocks), dim3(256), lds, st, p, gate, dx_add);
  return evt_check_launch();
}
