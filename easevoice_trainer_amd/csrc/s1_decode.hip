// s1 KV-cache decoding step (Text2SemanticDecoder.infer_panel_naive, t2s_model.py:762-863; T2SBlock.decode_next_token
// :186-222; sample / logits_to_probs, models/utils.py:118-160) for gfx950.
//
// One decoded token is a chain of batch-1 matrix-vector products over ~150 MB of weights plus attention over a growing
// key/value cache: HBM- and launch-bound, nothing for MFMA.  Laid out so that NOTHING about a step is a host argument:
//   * the key/value cache is preallocated [B][Lmax][E] and the current length lives in device memory (ctr[POS]) -- the
//     reference grows it with torch.cat every step;
//   * the step index, the number of generated tokens, the prompt length and the sampling seed are device state too
//     (ctr[IDX|YCOUNT|YLEN|SEED]);
//   * sampling (repetition penalty, nucleus cut, temperature, top-k, softmax, exponential-noise argmax), the EOS test,
//     the append to the token buffer and the embedding of the new token all run on the device.
// So the ~125 launches of a step are captured once into a HIP graph and replayed per token; the host only polls the stop
// flag every few steps.
//   dec_gemv   y = act(W . LN(a + r) + b): the post-LN residual of the previous sub-block is recomputed by every block in
//              its prologue (512 values) instead of being a launch of its own; block 0 stores it for the next residual.
//   dec_attn   one block per (batch, head): appends the new key/value to the cache, softmax(q.K/sqrt(d)).V over it.
//   dec_sample one block per batch row, 1024 threads, whole vocabulary (1025) in LDS, bitonic sort for the nucleus /
//              top-k pivots.
#include "evt_common.h"
#include "../../include/evt.h"

namespace {

constexpr int kMaxB = 4;

__device__ __forceinline__ float block_sum(float v, float* red, int nwaves) {
  v = wave_reduce_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float s = 0.f;
  for (int i = 0; i < nwaves; ++i) s += red[i];
  return s;
}

__device__ __forceinline__ float block_max(float v, float* red, int nwaves) {
  v = wave_reduce_max(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float s = -INFINITY;
  for (int i = 0; i < nwaves; ++i) s = fmaxf(s, red[i]);
  return s;
}

// (value, index) argmax with the FIRST index among equal values (torch.argmax on CPU); all threads get the result
__device__ __forceinline__ int block_argmax(float v, int i, float* redv, int* redi, int nwaves) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(v, o, 64);
    const int oi = __shfl_xor(i, o, 64);
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) { redv[w] = v; redi[w] = i; }
  __syncthreads();
  float bv = redv[0];
  int bi = redi[0];
  for (int k = 1; k < nwaves; ++k)
    if (redv[k] > bv || (redv[k] == bv && redi[k] < bi)) { bv = redv[k]; bi = redi[k]; }
  return bi;
}

template <typename T> struct WVec { static constexpr int V = 16 / sizeof(T); };

// ---- y[b][n] = act(bias[n] + sum_k W[n][k] * x[b][k]),  x = a  or  LayerNorm(a + r) --------------------------------
// A launch is a dependent chain of memory latencies, not a bandwidth problem (a workgroup touches 8-16 KB of weights), so
// the chain is kept as short as it gets: every weight load of the wave's rows is issued first (RPW x NPASS 16-byte loads
// in flight per lane), the input vector is staged -- and normalised with ONE block reduction of (sum, sum of squares) --
// while they are outstanding, then the dot products run out of registers and LDS.
template <typename T, int RPW, int NPASS>
__global__ __launch_bounds__(256) void dec_gemv(const T* __restrict__ W, const float* __restrict__ bias,
                                                const float* __restrict__ a, const float* __restrict__ r,
                                                const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                                float eps, float* x_out, float* __restrict__ y, int B, int N, int relu) {
  extern __shared__ float xs[];   // [B][K]
  __shared__ float red[8];
  constexpr int V = WVec<T>::V;
  constexpr int K = NPASS * 64 * V;
  constexpr int KPT = K / 256;     // input values per thread while staging
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = (blockIdx.x * 4 + wave) * RPW;
  uint4 w[RPW][NPASS];
#pragma unroll
  for (int i = 0; i < RPW; ++i)
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int n = n0 + i;
      w[i][ps] = n < N ? *reinterpret_cast<const uint4*>(W + (long)n * K + (ps * 64 + lane) * V) : make_uint4(0, 0, 0, 0);
    }
  // everything the epilogues need is requested now as well: a load placed after a barrier starts a new latency
  float lg[KPT], lb[KPT], bz[RPW];
#pragma unroll
  for (int e = 0; e < KPT; ++e) {
    lg[e] = r ? ln_g[tid + e * 256] : 1.f;
    lb[e] = r ? ln_b[tid + e * 256] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < RPW; ++i) bz[i] = (bias && n0 + i < N) ? bias[n0 + i] : 0.f;
  for (int b = 0; b < B; ++b) {
    float v[KPT];
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int e = 0; e < KPT; ++e) {
      const int k = tid + e * 256;
      v[e] = a[b * K + k] + (r ? r[b * K + k] : 0.f);
      s += v[e];
      q += v[e] * v[e];
    }
    if (r) {
      s = wave_reduce_sum(s);
      q = wave_reduce_sum(q);
      __syncthreads();
      if (lane == 0) { red[wave] = s; red[4 + wave] = q; }
      __syncthreads();
      const float mu = (red[0] + red[1] + red[2] + red[3]) / K;
      const float var = fmaxf((red[4] + red[5] + red[6] + red[7]) / K - mu * mu, 0.f);
      const float rs = rsqrtf(var + eps);
#pragma unroll
      for (int e = 0; e < KPT; ++e) {
        const int k = tid + e * 256;
        v[e] = (v[e] - mu) * rs * lg[e] + lb[e];
        if (x_out && blockIdx.x == 0) x_out[b * K + k] = v[e];
      }
    }
#pragma unroll
    for (int e = 0; e < KPT; ++e) xs[b * K + tid + e * 256] = v[e];
  }
  __syncthreads();
  float acc[RPW][kMaxB];
#pragma unroll
  for (int i = 0; i < RPW; ++i)
#pragma unroll
    for (int b = 0; b < kMaxB; ++b) acc[i][b] = 0.f;
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    const int k0 = (ps * 64 + lane) * V;
#pragma unroll
    for (int b = 0; b < kMaxB; ++b) {
      if (b >= B) break;
      float xv[V];
#pragma unroll
      for (int e = 0; e < V; ++e) xv[e] = xs[b * K + k0 + e];
#pragma unroll
      for (int i = 0; i < RPW; ++i) {
        const T* pw = reinterpret_cast<const T*>(&w[i][ps]);
#pragma unroll
        for (int e = 0; e < V; ++e) acc[i][b] += to_f<T>(pw[e]) * xv[e];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < RPW; ++i)
#pragma unroll
    for (int b = 0; b < kMaxB; ++b) {
      if (b >= B) break;
      const float sum = wave_reduce_sum(acc[i][b]);
      const int n = n0 + i;
      if (lane == 0 && n < N) {
        float o = sum + bz[i];
        if (relu) o = fmaxf(o, 0.f);
        y[(long)b * N + n] = o;
      }
    }
}

// ---- append (k, v) of the new token to the cache, attend over all cached positions -------------------------------
// Scores: one key per thread (D elements = D/V 16-byte loads, all in flight).  P.V: a thread owns one 16-byte chunk of
// the value rows of every G-th key (G = 256 / chunks-per-row), so the cache is read with 16-byte loads only; the G
// partial rows are summed through LDS.  The first key row and the first PF value chunks of a thread depend on nothing
// computed in the launch and are requested before anything else (Prefetch), the rest follows the softmax.
template <typename T, int D> struct AttnShape {
  static constexpr int V = WVec<T>::V;
  static constexpr int C = D / V;        // 16-byte chunks per row
  static constexpr int G = 256 / C;      // key groups in the P.V phase
  static constexpr int PF = 8;
};

template <typename T, int D> struct Prefetch {
  uint4 u0[AttnShape<T, D>::C], vpre[AttnShape<T, D>::PF];
  __device__ __forceinline__ void issue(const T* kc, const T* vc, int b, int h, int E, int Lmax, int L, int pos) {
    using S = AttnShape<T, D>;
    const int tid = threadIdx.x, g = tid / S::C, c = tid % S::C;
    if (tid < L && tid != pos) {
      const T* row = kc + ((long)b * Lmax + tid) * E + h * D;
#pragma unroll
      for (int cc = 0; cc < S::C; ++cc) u0[cc] = *reinterpret_cast<const uint4*>(row + cc * S::V);
    }
#pragma unroll
    for (int i = 0; i < S::PF; ++i) {
      const int j = g + i * S::G;
      if (j < L && j != pos) vpre[i] = *reinterpret_cast<const uint4*>(vc + ((long)b * Lmax + j) * E + h * D + c * S::V);
    }
  }
};

// qs (scaled query), kn / vn (new key / value, already rounded to the cache dtype) are in LDS and published
template <typename T, int D>
__device__ __forceinline__ void attn_tail(const Prefetch<T, D>& pf, const float* qs, const float* kn, const float* vn,
                                          float* sc, float* red, float (*part)[D + 1], const T* kc, const T* vc,
                                          float* __restrict__ out, int b, int h, int E, int Lmax, int L, int pos,
                                          int mlo, int mhi) {
  // keys mlo <= j < mhi are padding of a shorter text in a batch (infer_panel_batch_infer's padding mask): skipped
  using S = AttnShape<T, D>;
  constexpr int V = S::V, C = S::C, G = S::G, PF = S::PF;
  const int tid = threadIdx.x, g = tid / C, c = tid % C;
  float mx = -INFINITY;
  for (int j = tid; j < L; j += 256) {
    float s = 0.f;
    if (j >= mlo && j < mhi) {
      s = -INFINITY;
    } else if (j == pos) {
#pragma unroll
      for (int d = 0; d < D; ++d) s += qs[d] * kn[d];
    } else {
      uint4 u[C];
      if (j == tid) {
#pragma unroll
        for (int cc = 0; cc < C; ++cc) u[cc] = pf.u0[cc];
      } else {
        const T* row = kc + ((long)b * Lmax + j) * E + h * D;
#pragma unroll
        for (int cc = 0; cc < C; ++cc) u[cc] = *reinterpret_cast<const uint4*>(row + cc * V);
      }
#pragma unroll
      for (int cc = 0; cc < C; ++cc) {
        const T* pu = reinterpret_cast<const T*>(&u[cc]);
#pragma unroll
        for (int e = 0; e < V; ++e) s += qs[cc * V + e] * to_f<T>(pu[e]);
      }
    }
    sc[j] = s;
    mx = fmaxf(mx, s);
  }
  mx = block_max(mx, red, 4);
  float sum = 0.f;
  for (int j = tid; j < L; j += 256) {
    const float e = sc[j] == -INFINITY ? 0.f : expf(sc[j] - mx);
    sc[j] = e;
    sum += e;
  }
  sum = block_sum(sum, red, 4);      // its barriers also publish sc[]
  float acc[V];
#pragma unroll
  for (int e = 0; e < V; ++e) acc[e] = 0.f;
#pragma unroll
  for (int i = 0; i < PF; ++i) {
    const int j = g + i * G;
    if (j < L) {
      const float pj = sc[j];
      if (j == pos) {
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] += pj * vn[c * V + e];
      } else {
        const T* pu = reinterpret_cast<const T*>(&pf.vpre[i]);
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] += pj * to_f<T>(pu[e]);
      }
    }
  }
  for (int j = g + PF * G; j < L; j += G) {
    const float pj = sc[j];
    if (j == pos) {
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] += pj * vn[c * V + e];
    } else {
      const uint4 u = *reinterpret_cast<const uint4*>(vc + ((long)b * Lmax + j) * E + h * D + c * V);
      const T* pu = reinterpret_cast<const T*>(&u);
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] += pj * to_f<T>(pu[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < V; ++e) part[g][c * V + e] = acc[e];
  __syncthreads();
  if (tid < D) {
    float o = 0.f;
#pragma unroll 8
    for (int i = 0; i < G; ++i) o += part[i][tid];
    out[(long)b * E + h * D + tid] = o / sum;
  }
}

template <typename T, int D>
__global__ __launch_bounds__(256) void dec_attn(const float* __restrict__ qkv, T* kc, T* vc, const int* __restrict__ ctr,
                                                float* __restrict__ out, int H, int Lmax,
                                                const int* __restrict__ x_lens, int x_len) {
  extern __shared__ float sc[];   // [Lmax] scores -> probabilities
  __shared__ float qs[D], kn[D], vn[D], red[4], part[AttnShape<T, D>::G][D + 1];
  const int tid = threadIdx.x;
  const int b = blockIdx.x / H, h = blockIdx.x % H, E = H * D;
  const int pos = ctr[EVT_DEC_POS];
  if (pos >= Lmax) return;            // cache full: the host bounds the number of steps, this only guards memory
  const int L = pos + 1;
  Prefetch<T, D> pf;
  pf.issue(kc, vc, b, h, E, Lmax, L, pos);
  if (tid < D) {
    const float* base = qkv + (long)b * 3 * E + h * D + tid;
    qs[tid] = base[0] * rsqrtf((float)D);
    const T kq = from_f<T>(base[E]), vq = from_f<T>(base[2 * E]);
    kn[tid] = to_f<T>(kq);
    vn[tid] = to_f<T>(vq);
    kc[((long)b * Lmax + pos) * E + h * D + tid] = kq;
    vc[((long)b * Lmax + pos) * E + h * D + tid] = vq;
  }
  __syncthreads();
  attn_tail<T, D>(pf, qs, kn, vn, sc, red, part, kc, vc, out, b, h, E, Lmax, L, pos, x_lens ? x_lens[b] : 0,
                  x_lens ? x_len : 0);
}

// ---- the same with the head's own rows of the packed in-projection computed in the launch -------------------------
// A workgroup (batch b, head h) needs only rows {q, k, v} x [h*D, (h+1)*D) of W_qkv: 3*D rows of E values (96 KB in
// bf16).  Computing them here removes the qkv launch of every block from the token step.  Input x = a or
// LayerNorm(a + r) as in dec_gemv (workgroups h == 0 store it to x_out).
template <typename T, int D, int NPASS>
__global__ __launch_bounds__(256) void dec_qkv_attn(const T* __restrict__ W, const float* __restrict__ bias,
                                                    const float* __restrict__ a, const float* __restrict__ r,
                                                    const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                                    float eps, float* x_out, T* kc, T* vc, const int* __restrict__ ctr,
                                                    float* __restrict__ out, int H, int Lmax,
                                                    const int* __restrict__ x_lens, int x_len) {
  extern __shared__ float sc[];   // [Lmax]
  constexpr int V = WVec<T>::V;
  constexpr int E = NPASS * 64 * V;       // model width == K of the projection
  constexpr int KPT = E / 256;
  constexpr int ROWS = 3 * D, RPWV = ROWS / 4;   // rows per wave
  constexpr int RC = NPASS == 1 ? RPWV : RPWV / 2;   // rows whose weights are in flight at once
  __shared__ float xs[E], proj[ROWS], qs[D], kn[D], vn[D], red[8], part[AttnShape<T, D>::G][D + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int pos = ctr[EVT_DEC_POS];
  if (pos >= Lmax) return;
  const int L = pos + 1;
  // projection row rho in [0, 3D): part = rho / D selects q / k / v, global row = part*E + h*D + rho % D
  auto grow = [&](int rho) { return (rho / D) * E + h * D + rho % D; };
  uint4 w[RC][NPASS];
#pragma unroll
  for (int i = 0; i < RC; ++i)
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps)
      w[i][ps] = *reinterpret_cast<const uint4*>(W + (long)grow(wave * RPWV + i) * E + (ps * 64 + lane) * V);
  Prefetch<T, D> pf;
  pf.issue(kc, vc, b, h, E, Lmax, L, pos);
  float lg[KPT], lb[KPT], v[KPT];
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int e = 0; e < KPT; ++e) {
    const int k = tid + e * 256;
    lg[e] = r ? ln_g[k] : 1.f;
    lb[e] = r ? ln_b[k] : 0.f;
    v[e] = a[b * E + k] + (r ? r[b * E + k] : 0.f);
    s += v[e];
    q += v[e] * v[e];
  }
  const float bz = tid < ROWS ? bias[grow(tid)] : 0.f;
  if (r) {
    s = wave_reduce_sum(s);
    q = wave_reduce_sum(q);
    if (lane == 0) { red[wave] = s; red[4 + wave] = q; }
    __syncthreads();
    const float mu = (red[0] + red[1] + red[2] + red[3]) / E;
    const float var = fmaxf((red[4] + red[5] + red[6] + red[7]) / E - mu * mu, 0.f);
    const float rs = rsqrtf(var + eps);
#pragma unroll
    for (int e = 0; e < KPT; ++e) {
      v[e] = (v[e] - mu) * rs * lg[e] + lb[e];
      if (x_out && h == 0) x_out[b * E + tid + e * 256] = v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < KPT; ++e) xs[tid + e * 256] = v[e];
  __syncthreads();
#pragma unroll
  for (int chunk = 0; chunk < RPWV / RC; ++chunk) {
    if (chunk > 0) {
#pragma unroll
      for (int i = 0; i < RC; ++i)
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps)
          w[i][ps] = *reinterpret_cast<const uint4*>(W + (long)grow(wave * RPWV + chunk * RC + i) * E + (ps * 64 + lane) * V);
    }
#pragma unroll
    for (int i = 0; i < RC; ++i) {
      float acc = 0.f;
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        const T* pw = reinterpret_cast<const T*>(&w[i][ps]);
        const int k0 = (ps * 64 + lane) * V;
#pragma unroll
        for (int e = 0; e < V; ++e) acc += to_f<T>(pw[e]) * xs[k0 + e];
      }
      acc = wave_reduce_sum(acc);
      if (lane == 0) proj[wave * RPWV + chunk * RC + i] = acc;
    }
  }
  __syncthreads();
  if (tid < ROWS) proj[tid] += bz;
  __syncthreads();
  if (tid < D) {
    qs[tid] = proj[tid] * rsqrtf((float)D);
    const T kq = from_f<T>(proj[D + tid]), vq = from_f<T>(proj[2 * D + tid]);
    kn[tid] = to_f<T>(kq);
    vn[tid] = to_f<T>(vq);
    kc[((long)b * Lmax + pos) * E + h * D + tid] = kq;
    vc[((long)b * Lmax + pos) * E + h * D + tid] = vq;
  }
  __syncthreads();
  attn_tail<T, D>(pf, qs, kn, vn, sc, red, part, kc, vc, out, b, h, E, Lmax, L, pos, x_lens ? x_lens[b] : 0,
                  x_lens ? x_len : 0);
}

// ---- sampling ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned mix32s(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

__device__ __forceinline__ bool before(float av, int ai, float bv, int bi) {   // descending value, ascending index
  return av > bv || (av == bv && ai < bi);
}

constexpr int kSortN = 2048;

struct EmbedArgs {          // optional tail of dec_sample: x_next and the counter update of the step (B == 1 only)
  const float* emb; const float* pe; const float* alpha; float* x; float x_scale; int E, npos, dpos;
};

__global__ __launch_bounds__(1024) void dec_sample(evt_sample_params p, const float* __restrict__ logits, long* y,
                                                   int* ctr, const float* __restrict__ noise, int* stop_idx,
                                                   float* probs_out, EmbedArgs ea) {
  __shared__ float sv[kSortN];
  __shared__ int si[kSortN];
  __shared__ float cur[kSortN];
  __shared__ unsigned char flag[kSortN];
  __shared__ float redv[16];
  __shared__ int redi[16];
  __shared__ float wsum[16];
  const int tid = threadIdx.x, b = blockIdx.x, V = p.V;
  const int idx = ctr[EVT_DEC_IDX], ycount = ctr[EVT_DEC_YCOUNT];
  const int Ve = idx < p.no_eos_steps ? V - 1 : V;     // "at least 10 tokens otherwise not stop", t2s_model.py:833
  const float* lg = logits + (long)b * V;
  long* yb = y + (long)b * p.ymax;
  for (int v = tid; v < kSortN; v += 1024) flag[v] = 0;
  __syncthreads();
  if (p.repetition_penalty != 1.0f)
    for (int j = tid; j < ycount; j += 1024) {
      const long t = yb[j];
      if (t >= 0 && t < Ve) flag[t] = 1;
    }
  __syncthreads();
  float bvv = -INFINITY;
  int bii = 0x7fffffff;
  for (int v = tid; v < kSortN; v += 1024) {
    float x = -INFINITY;
    if (v < Ve) {
      x = lg[v];
      if (flag[v]) x = x < 0.f ? x * p.repetition_penalty : x / p.repetition_penalty;
      if (x > bvv || (x == bvv && v < bii)) { bvv = x; bii = v; }
    }
    cur[v] = x;
    sv[v] = x;
    si[v] = v;
  }
  // argmax of the (penalised, in place in the reference) logits: the EOS test of t2s_model.py:846
  const int amax = block_argmax(bvv, bii, redv, redi, 16);
  // bitonic sort, descending
  for (int k = 2; k <= kSortN; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      const int i = 2 * j * (tid / j) + (tid % j), l = i + j;
      const bool up = (i & k) == 0;
      const float av = sv[i], bv = sv[l];
      const int ai = si[i], bi = si[l];
      const bool in_order = before(av, ai, bv, bi);
      if (in_order != up) { sv[i] = bv; sv[l] = av; si[i] = bi; si[l] = ai; }
    }
  __syncthreads();
  if (p.top_p < 1.0f) {
    // cumulative softmax over the sorted logits; entries past the nucleus are removed, the first is always kept
    const float m = sv[0];
    const float e0 = expf(sv[2 * tid] - m), e1 = expf(sv[2 * tid + 1] - m);
    float run = e0 + e1;
    const int lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const float t = __shfl_up(run, o, 64);
      if (lane >= o) run += t;
    }
    if (lane == 63) wsum[w] = run;
    __syncthreads();
    float offs = 0.f, total = 0.f;
    for (int i = 0; i < 16; ++i) {
      if (i < w) offs += wsum[i];
      total += wsum[i];
    }
    const float c1 = (offs + run) / total, c0 = (offs + run - e1) / total;
    if (2 * tid > 0 && c0 > p.top_p && si[2 * tid] < Ve) cur[si[2 * tid]] = -INFINITY;
    if (c1 > p.top_p && si[2 * tid + 1] < Ve) cur[si[2 * tid + 1]] = -INFINITY;
    __syncthreads();
  }
  const float tdiv = fmaxf(p.temperature, 1e-5f);
  float pivot = -INFINITY;
  if (p.top_k > 0) {
    const int kk = p.top_k < Ve ? p.top_k : Ve;
    pivot = cur[si[kk - 1]] / tdiv;
  }
  float x0[2], mx = -INFINITY;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int v = tid + u * 1024;
    float x = -INFINITY;
    if (v < Ve) {
      x = cur[v] / tdiv;
      if (x < pivot) x = -INFINITY;
    }
    x0[u] = x;
    mx = fmaxf(mx, x);
  }
  mx = block_max(mx, redv, 16);
  float e[2], sum = 0.f;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    e[u] = x0[u] == -INFINITY ? 0.f : expf(x0[u] - mx);
    sum += e[u];
  }
  sum = block_sum(sum, redv, 16);
  float best = -INFINITY;
  int besti = 0x7fffffff;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int v = tid + u * 1024;
    if (v < Ve) {
      const float pr = e[u] / sum;
      if (probs_out) probs_out[(long)b * V + v] = pr;
      float q;
      if (noise) {
        q = noise[((long)idx * p.noise_rows + (p.noise_rows > 1 ? b : 0)) * V + v];
      } else {
        const unsigned hsh =
            mix32s(mix32s((p.seed ^ (unsigned)ctr[EVT_DEC_SEED]) + (unsigned)idx * 0x9E3779B9u) ^ ((unsigned)b << 16) ^ (unsigned)v);
        q = -logf(((float)(hsh >> 8) + 0.5f) * (1.0f / 16777216.0f));
      }
      const float s = pr / q;
      if (s > best || (s == best && v < besti)) { best = s; besti = v; }
    } else if (probs_out && v < V) {
      probs_out[(long)b * V + v] = 0.f;
    }
  }
  const int tok = block_argmax(best, besti, redv, redi, 16);
  if (tid == 0) {
    if (ycount < p.ymax) yb[ycount] = tok;
    if ((amax == p.eos || tok == p.eos) && stop_idx[b] < 0) stop_idx[b] = idx;
  }
  if (ea.x) {       // x_next = emb[token] * x_scale + alpha * pe[y_len + idx], then the step's counter update
    int ppos = ctr[EVT_DEC_YLEN] + idx;
    if (ppos >= ea.npos) ppos = ea.npos - 1;
    const float al = ea.alpha[0];
    for (int c = tid; c < ea.E; c += 1024)
      ea.x[(long)b * ea.E + c] = ea.emb[(long)tok * ea.E + c] * ea.x_scale + al * ea.pe[(long)ppos * ea.E + c];
    __syncthreads();     // every read of ctr[] above is done (single workgroup: the launcher checks B == 1)
    if (tid == 0) {
      ctr[EVT_DEC_POS] += ea.dpos;
      ctr[EVT_DEC_IDX] = idx + 1;
      ctr[EVT_DEC_YCOUNT] = ycount + 1;
    }
  }
}

// ---- x_next = emb[token] * x_scale + alpha * pe[y_len + idx]  (t2s_model.py:860-861) ----------------------------
__global__ __launch_bounds__(256) void dec_embed(const float* __restrict__ emb, const float* __restrict__ pe,
                                                 const float* __restrict__ alpha, float x_scale,
                                                 const long* __restrict__ y, const int* __restrict__ ctr,
                                                 float* __restrict__ x, int E, int ymax, int npos) {
  const int b = blockIdx.x;
  const int idx = ctr[EVT_DEC_IDX], ycount = ctr[EVT_DEC_YCOUNT], ylen = ctr[EVT_DEC_YLEN];
  const long tok = y[(long)b * ymax + (ycount < ymax ? ycount : ymax - 1)];
  int ppos = ylen + idx;
  if (ppos >= npos) ppos = npos - 1;
  const float al = alpha[0];
  for (int c = threadIdx.x; c < E; c += 256) x[(long)b * E + c] = emb[tok * E + c] * x_scale + al * pe[(long)ppos * E + c];
}

__global__ void dec_advance(int* ctr, int dpos) {
  ctr[EVT_DEC_POS] += dpos;
  ctr[EVT_DEC_IDX] += 1;
  ctr[EVT_DEC_YCOUNT] += 1;
}

}  // namespace

template <typename T>
static int launch_gemv(const void* W, const float* bias, const float* a, const float* r, const float* g, const float* bt,
                       float eps, float* x_out, float* y, int B, int N, int K, int relu, hipStream_t st) {
  constexpr int V = WVec<T>::V;
  const size_t shm = (size_t)B * K * sizeof(float);
  const int npass = K / (64 * V);
  // rows per wave: keep >= ~128 workgroups in flight for the short matrices, two rows per wave for the tall ones
  const int rpw = N >= 1024 ? 2 : 1;
  const int blocks = (N + 4 * rpw - 1) / (4 * rpw);
#define EVT_GEMV(RPW, NP)                                                                                            \
  hipLaunchKernelGGL((dec_gemv<T, RPW, NP>), dim3(blocks), dim3(256), shm, st, (const T*)W, bias, a, r, g, bt, eps, \
                     x_out, y, B, N, relu)
  if (npass * 64 * V != K) return EVT_ENOTSUP;
  if (rpw == 2) {
    if (npass == 1) EVT_GEMV(2, 1); else if (npass == 2) EVT_GEMV(2, 2); else if (npass == 4) EVT_GEMV(2, 4);
    else if (npass == 8) EVT_GEMV(2, 8); else return EVT_ENOTSUP;
  } else {
    if (npass == 1) EVT_GEMV(1, 1); else if (npass == 2) EVT_GEMV(1, 2); else if (npass == 4) EVT_GEMV(1, 4);
    else if (npass == 8) EVT_GEMV(1, 8); else return EVT_ENOTSUP;
  }
#undef EVT_GEMV
  return evt_check_launch();
}

extern "C" {

int evt_dec_gemv(int32_t wdtype, const void* W, const float* bias, const float* a, const float* r, const float* ln_g,
                 const float* ln_b, float ln_eps, float* x_out, float* y, int32_t B, int32_t N, int32_t K, int32_t relu,
                 void* stream) {
  if (!W || !a || !y || B <= 0 || N <= 0 || K <= 0) return EVT_EINVAL;
  if (r && (!ln_g || !ln_b)) return EVT_EINVAL;
  if (B > kMaxB || K % 512 || (size_t)B * K * 4 > 64 * 1024) return EVT_ENOTSUP;
  hipStream_t st = (hipStream_t)stream;
  if (wdtype == EVT_DT_HALF) return launch_gemv<h16_t>(W, bias, a, r, ln_g, ln_b, ln_eps, x_out, y, B, N, K, relu, st);
  if (wdtype == EVT_DT_F32) return launch_gemv<float>(W, bias, a, r, ln_g, ln_b, ln_eps, x_out, y, B, N, K, relu, st);
  return EVT_EINVAL;
}

int evt_dec_attn(int32_t cdtype, const float* qkv, void* kcache, void* vcache, const int32_t* ctr, float* out, int32_t B,
                 int32_t H, int32_t D, int32_t Lmax, const int32_t* x_lens, int32_t x_len, void* stream) {
  if (!qkv || !kcache || !vcache || !ctr || !out || B <= 0 || H <= 0 || Lmax <= 0) return EVT_EINVAL;
  if (D != 32 || (size_t)Lmax * 4 > 60 * 1024) return EVT_ENOTSUP;
  const size_t shm = (size_t)Lmax * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (cdtype == EVT_DT_HALF)
    hipLaunchKernelGGL((dec_attn<h16_t, 32>), dim3(B * H), dim3(256), shm, st, qkv, (h16_t*)kcache, (h16_t*)vcache,
                       (const int*)ctr, out, H, Lmax, (const int*)x_lens, x_len);
  else if (cdtype == EVT_DT_F32)
    hipLaunchKernelGGL((dec_attn<float, 32>), dim3(B * H), dim3(256), shm, st, qkv, (float*)kcache, (float*)vcache,
                       (const int*)ctr, out, H, Lmax, (const int*)x_lens, x_len);
  else return EVT_EINVAL;
  return evt_check_launch();
}

int evt_dec_sample(const evt_sample_params* p, const float* logits, int64_t* y, const int32_t* ctr, const float* noise,
                   int32_t* stop_idx, float* probs_out, int32_t B, void* stream) {
  if (!p || !logits || !y || !ctr || !stop_idx || B <= 0) return EVT_EINVAL;
  if (p->V <= 1 || p->V > kSortN || p->ymax <= 0 || p->repetition_penalty <= 0.f) return EVT_EINVAL;
  EmbedArgs none{};
  evt_sample_params sp = *p;
  if (sp.noise_rows < 1) sp.noise_rows = 1;
  hipLaunchKernelGGL(dec_sample, dim3(B), dim3(1024), 0, (hipStream_t)stream, sp, logits, (long*)y, (int*)ctr, noise,
                     (int*)stop_idx, probs_out, none);
  return evt_check_launch();
}

int evt_dec_sample_embed(const evt_sample_params* p, const float* logits, int64_t* y, int32_t* ctr, const float* noise,
                         int32_t* stop_idx, const float* emb, const float* pe, const float* alpha, float x_scale, float* x,
                         int32_t E, int32_t npos, int32_t dpos, void* stream) {
  if (!p || !logits || !y || !ctr || !stop_idx || !emb || !pe || !alpha || !x || E <= 0 || npos <= 0) return EVT_EINVAL;
  if (p->V <= 1 || p->V > kSortN || p->ymax <= 0 || p->repetition_penalty <= 0.f) return EVT_EINVAL;
  EmbedArgs ea{emb, pe, alpha, x, x_scale, E, npos, dpos};
  evt_sample_params sp = *p;
  sp.noise_rows = 1;
  hipLaunchKernelGGL(dec_sample, dim3(1), dim3(1024), 0, (hipStream_t)stream, sp, logits, (long*)y, (int*)ctr, noise,
                     (int*)stop_idx, (float*)nullptr, ea);
  return evt_check_launch();
}

int evt_dec_qkv_attn(int32_t dtype, const void* Wqkv, const float* bqkv, const float* a, const float* r,
                     const float* ln_g, const float* ln_b, float ln_eps, float* x_out, void* kcache, void* vcache,
                     const int32_t* ctr, float* out, int32_t B, int32_t H, int32_t D, int32_t Lmax, const int32_t* x_lens,
                     int32_t x_len, void* stream) {
  if (!Wqkv || !bqkv || !a || !kcache || !vcache || !ctr || !out || B <= 0 || H <= 0 || Lmax <= 0) return EVT_EINVAL;
  if (r && (!ln_g || !ln_b)) return EVT_EINVAL;
  if (D != 32 || H * D != 512 || (size_t)Lmax * 4 > 48 * 1024) return EVT_ENOTSUP;
  const size_t shm = (size_t)Lmax * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EVT_DT_HALF)
    hipLaunchKernelGGL((dec_qkv_attn<h16_t, 32, 1>), dim3(B * H), dim3(256), shm, st, (const h16_t*)Wqkv, bqkv, a, r,
                       ln_g, ln_b, ln_eps, x_out, (h16_t*)kcache, (h16_t*)vcache, (const int*)ctr, out, H, Lmax,
                       (const int*)x_lens, x_len);
  else if (dtype == EVT_DT_F32)
    hipLaunchKernelGGL((dec_qkv_attn<float, 32, 2>), dim3(B * H), dim3(256), shm, st, (const float*)Wqkv, bqkv, a, r,
                       ln_g, ln_b, ln_eps, x_out, (float*)kcache, (float*)vcache, (const int*)ctr, out, H, Lmax,
                       (const int*)x_lens, x_len);
  else return EVT_EINVAL;
  return evt_check_launch();
}

int evt_dec_embed(const float* emb, const float* pe, const float* alpha, float x_scale, const int64_t* y,
                  const int32_t* ctr, float* x, int32_t B, int32_t E, int32_t ymax, int32_t npos, void* stream) {
  if (!emb || !pe || !alpha || !y || !ctr || !x || B <= 0 || E <= 0 || ymax <= 0 || npos <= 0) return EVT_EINVAL;
  hipLaunchKernelGGL(dec_embed, dim3(B), dim3(256), 0, (hipStream_t)stream, emb, pe, alpha, x_scale, (const long*)y,
                     (const int*)ctr, x, E, ymax, npos);
  return evt_check_launch();
}

int evt_dec_advance(int32_t* ctr, int32_t dpos, void* stream) {
  if (!ctr) return EVT_EINVAL;
  hipLaunchKernelGGL(dec_advance, dim3(1), dim3(1), 0, (hipStream_t)stream, (int*)ctr, dpos);
  return evt_check_launch();
}

}  // extern "C"
