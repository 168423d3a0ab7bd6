// Degenerate Conv1d shapes of the s2 path that have no GEMM in them (gfx950):
//   * Cout == 1  (HiFi-GAN conv_post 16->1 k7, models.py:446; discriminator conv_post 1024->1 k3, models.py:536,574):
//     a dot product per output position  -> lane groups reduce over (tap, channel) with 16-byte loads;
//     its weight gradient is a dy-weighted sum of input rows -> per-thread register accumulators.
//   * Cin == 1   (discriminator first layers 1->16 k15 / 1->32 k5 s3, models.py:490-497,566): weight gradient as
//     per-(channel, tap) register accumulators over positions.
// HBM-bound byte work: coalesced 16-byte reads, LDS only for the small weight vector / block reduction.
#include "evt_common.h"
#include "../../include/evt.h"

namespace {

struct SP {
  const void* x; const void* w; const float* bias; const void* y_in; const void* dy; void* y; float* dw;
  int nseq, lin, lout, cin, cout, k, stride, pad, dil;
  int ck, nchunk, kp;   // REG geometry
  float in_slope; int out_act; float out_slope;
  int G;                // lanes per output (power of two <= 64)
  int pos_per_block;
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
  float* red = reinterpret_cast<float*>(smem);   // [k*cin]
  const int ppr = p.cin / V;
  const int pieces = p.k * ppr;
  int pp = 1;
  while (pp < pieces && pp < 256) pp <<= 1;       // pieces padded to a power of two (<= 256)
  const int npl = 256 / pp;                        // positions processed in parallel
  const int pl = threadIdx.x / pp, pc0 = threadIdx.x % pp;
  for (int i = threadIdx.x; i < p.k * p.cin; i += 256) red[i] = 0.f;
  __syncthreads();
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
  for (long o = p0 + pl; o < p1; o += npl) {
    const int q = (int)(o % p.lout), seq = (int)(o / p.lout);
    float d = to_f<T>(dy[o]);
    if (ys) d *= dact_from_out(p.out_act, to_f<T>(ys[o]), p.out_slope);
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      const int pc = pc0 + a * pp;
      if (pc >= pieces) continue;
      const int t = pc / ppr, c0 = (pc - t * ppr) * V;
      const int row = q * p.stride + t * p.dil - p.pad;
      if (row < 0 || row >= p.lin) continue;
      const uint4 v = *reinterpret_cast<const uint4*>(x + ((long)seq * p.lin + row) * p.cin + c0);
      const T* pv = reinterpret_cast<const T*>(&v);
#pragma unroll
      for (int e = 0; e < V; ++e) acc[a][e] += d * lrelu_f(to_f<T>(pv[e]), p.in_slope);
    }
  }
#pragma unroll
  for (int a = 0; a < NA; ++a) {
    const int pc = pc0 + a * pp;
    if (pc < pieces) {
      const int t = pc / ppr, c0 = (pc - t * ppr) * V;
#pragma unroll
      for (int e = 0; e < V; ++e) atomicAdd(&red[t * p.cin + c0 + e], acc[a][e]);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < p.k * p.cin; i += 256) {
    const int t = i / p.cin, c = i - t * p.cin;
    atomicAdd(p.dw + sreg_index(p, 0, c, t), red[i]);
  }
}

// ---- Cin == 1 backward-weight: dW[co][t] += sum_pos dy_eff[pos][co] * x[pos*s + t*dil - pad] ------------------
template <typename T, int KMAX>
__global__ __launch_bounds__(256) void cin1_bwd_weight(SP p) {
  __shared__ float red[64 * KMAX];   // cout <= 64
  const int co = threadIdx.x % p.cout, pl = threadIdx.x / p.cout;
  const int npl = 256 / p.cout;
  for (int i = threadIdx.x; i < p.cout * p.k; i += 256) red[i] = 0.f;
  __syncthreads();
  float acc[KMAX];
#pragma unroll
  for (int t = 0; t < KMAX; ++t) acc[t] = 0.f;
  const T* x = reinterpret_cast<const T*>(p.x);
  const T* dy = reinterpret_cast<const T*>(p.dy);
  const T* ys = reinterpret_cast<const T*>(p.y_in);
  const long total = (long)p.nseq * p.lout;
  const long p0 = (long)blockIdx.x * p.pos_per_block;
  const long p1 = min(total, p0 + p.pos_per_block);
  if (pl < npl) {
    for (long o = p0 + pl; o < p1; o += npl) {
      const int q = (int)(o % p.lout), seq = (int)(o / p.lout);
      float d = to_f<T>(dy[o * p.cout + co]);
      if (ys) d *= dact_from_out(p.out_act, to_f<T>(ys[o * p.cout + co]), p.out_slope);
      const T* xr = x + (long)seq * p.lin;
#pragma unroll
      for (int t = 0; t < KMAX; ++t) {
        if (t < p.k) {
          const int row = q * p.stride + t * p.dil - p.pad;
          if (row >= 0 && row < p.lin) acc[t] += d * lrelu_f(to_f<T>(xr[row]), p.in_slope);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < KMAX; ++t)
      if (t < p.k) atomicAdd(&red[co * p.k + t], acc[t]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < p.cout * p.k; i += 256) {
    const int c = i / p.k, t = i - c * p.k;
    atomicAdd(p.dw + sreg_index(p, c, 0, t), red[i]);
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

}  // namespace

extern "C" int evt_small_kind(const evt_conv1d_params* c) {
  if (c->transposed || c->groups != 1) return 0;
  const int V = c->dtype == EVT_DT_BF16 ? 8 : 4;
  if (c->cout == 1 && c->cin % V == 0 && (long)c->k * c->cin <= 4096) return 1;   // dot-product conv
  if (c->cin == 1 && c->cout <= 64 && 256 % c->cout == 0 && c->k <= 16) return 2;  // single-channel input
  return 0;
}

extern "C" int evt_cout1_fwd(const evt_conv1d_params* c, const void* x, const void* w_reg, const float* bias, void* y,
                             void* stream) {
  SP p = make_sp(c);
  p.x = x; p.w = w_reg; p.bias = bias; p.y = y;
  const int V = c->dtype == EVT_DT_BF16 ? 8 : 4;
  const int pieces = c->k * (c->cin / V);
  int G = 1;
  while (G < pieces && G < 64) G <<= 1;
  p.G = G;
  const long total = (long)p.nseq * p.lout;
  long blocks = (total * G + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  const size_t lds = (size_t)c->k * c->cin * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (c->dtype == EVT_DT_BF16) hipLaunchKernelGGL(cout1_fwd<bf16_t>, dim3((int)blocks), dim3(256), lds, st, p);
  else hipLaunchKernelGGL(cout1_fwd<float>, dim3((int)blocks), dim3(256), lds, st, p);
  return evt_check_launch();
}

extern "C" int evt_cout1_bwd_weight(const evt_conv1d_params* c, const void* x, const void* dy, const void* y, float* dw,
                                    void* stream) {
  SP p = make_sp(c);
  p.x = x; p.dy = dy; p.y_in = c->out_act != EVT_ACT_NONE ? y : nullptr; p.dw = dw;
  const long total = (long)p.nseq * p.lout;
  long ppb = (total + 511) / 512;
  if (ppb < 64) ppb = 64;
  p.pos_per_block = (int)ppb;
  const int blocks = (int)((total + ppb - 1) / ppb);
  const size_t lds = (size_t)c->k * c->cin * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (c->dtype == EVT_DT_BF16) hipLaunchKernelGGL(cout1_bwd_weight<bf16_t>, dim3(blocks), dim3(256), lds, st, p);
  else hipLaunchKernelGGL(cout1_bwd_weight<float>, dim3(blocks), dim3(256), lds, st, p);
  return evt_check_launch();
}

extern "C" int evt_cin1_bwd_weight(const evt_conv1d_params* c, const void* x, const void* dy, const void* y, float* dw,
                                   void* stream) {
  SP p = make_sp(c);
  p.x = x; p.dy = dy; p.y_in = c->out_act != EVT_ACT_NONE ? y : nullptr; p.dw = dw;
  const long total = (long)p.nseq * p.lout;
  long ppb = (total + 1023) / 1024;
  if (ppb < 64) ppb = 64;
  p.pos_per_block = (int)ppb;
  const int blocks = (int)((total + ppb - 1) / ppb);
  hipStream_t st = (hipStream_t)stream;
  if (c->dtype == EVT_DT_BF16) hipLaunchKernelGGL((cin1_bwd_weight<bf16_t, 16>), dim3(blocks), dim3(256), 0, st, p);
  else hipLaunchKernelGGL((cin1_bwd_weight<float, 16>), dim3(blocks), dim3(256), 0, st, p);
  return evt_check_launch();
}
