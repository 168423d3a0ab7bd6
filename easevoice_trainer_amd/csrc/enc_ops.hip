// Fused glue of the s2 relative-position transformer encoders (enc_p: encoder_ssl / encoder_text / encoder2), gfx950.
//
// Reference call sites (file:line under /root/reference), src/easevoice/module/attentions.py:
//   :60-75   y = drop(attn(x)); x = norm_1(x + y); y = drop(ffn(x)); x = norm_2(x + y)      -> res_drop_ln_fwd/bwd
//   modules.py:19-31 LayerNorm over channels (gamma, beta)
// The reference issues dropout, add, (autocast casts,) layer_norm and the x * x_mask of the next consumer as separate
// element-wise launches over a [B, T, 192] tensor -- ~3 us of launch each for ~1 us of HBM traffic.  Here one launch
// reads x and y once, and writes LayerNorm(x + dropout(y)) * row_mask in the compute dtype; the backward regenerates
// the dropout mask from the counter hash instead of storing it.
//
// Row mask: row (b, t) is live iff t < lens[b].  Masked rows are written as zeros.  (The reference leaves LayerNorm(beta)
// garbage in masked rows and multiplies by x_mask at every consumer; masked rows never reach a live row -- attention masks
// them as keys, the FFN convolutions take x * x_mask -- so live rows and all parameter gradients are unchanged.)
//
// Dropout: keep(element) = hash(*seed_dev, site, element index) >= p * 2^32, scaled by 1/(1-p).  *seed_dev is a device
// counter the engine bumps once per step (evt_counter_inc), so a replayed HIP graph draws fresh masks every step.
#include "evt_common.h"
#include <type_traits>
#include <cstdlib>
#include "../../include/evt.h"

namespace {

__device__ __forceinline__ unsigned mix32(unsigned x) {   // lowbias32 finaliser
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

struct DropCfg {
  unsigned key;      // mix of *seed_dev and the site id
  unsigned thr;      // keep iff hash >= thr; 0 = dropout off
  float scale;       // 1 / (1 - p)
};

__device__ __forceinline__ DropCfg drop_cfg(float p, const unsigned* seed_dev, unsigned site) {
  DropCfg d;
  d.thr = p > 0.f ? (unsigned)fminf(p * 4294967296.f, 4294967040.f) : 0u;
  d.scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
  d.key = mix32((seed_dev ? *seed_dev : 0u) * 0x9E3779B1u + site * 0x85EBCA77u + 0x165667B1u);
  return d;
}

__device__ __forceinline__ float drop_mult(const DropCfg& d, unsigned long idx) {
  if (d.thr == 0u) return 1.f;
  const unsigned h = mix32((unsigned)idx ^ d.key ^ (unsigned)(idx >> 32) * 0xC2B2AE35u);
  return h >= d.thr ? d.scale : 0.f;
}

template <typename T> struct Vec16 { static constexpr int V = 16 / sizeof(T); };

// one wave per row, 16-byte accesses; a lane owns V consecutive channels per pass (C % V == 0, C <= 64*V*NP)
template <typename T, int NP>
__global__ __launch_bounds__(256) void res_drop_ln_fwd(const T* x, const T* y, const float* gamma, const float* beta,
                                                       const int* lens, int rows_per_seq, float p,
                                                       const unsigned* seed_dev, unsigned site, T* out, float* mean,
                                                       float* rstd, long rows, int C, float eps) {
  constexpr int V = Vec16<T>::V;
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  bool live = true;
  if (lens) { const long b = row / rows_per_seq; live = (int)(row - b * rows_per_seq) < lens[b]; }
  if (!live) {   // wave-uniform
#pragma unroll
    for (int pss = 0; pss < NP; ++pss) {
      const int c0 = (pss * 64 + lane) * V;
      if (c0 < C) *reinterpret_cast<uint4*>(out + row * C + c0) = make_uint4(0, 0, 0, 0);
    }
    if (lane == 0) { mean[row] = 0.f; rstd[row] = 0.f; }
    return;
  }
  const DropCfg dc = drop_cfg(p, seed_dev, site);
  float v[NP][V];
  float s = 0.f;
#pragma unroll
  for (int pss = 0; pss < NP; ++pss) {
    const int c0 = (pss * 64 + lane) * V;
    if (c0 < C) {
      const uint4 a = *reinterpret_cast<const uint4*>(x + row * C + c0);
      const uint4 b = *reinterpret_cast<const uint4*>(y + row * C + c0);
      const T* pa = reinterpret_cast<const T*>(&a);
      const T* pb = reinterpret_cast<const T*>(&b);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        v[pss][e] = to_f<T>(pa[e]) + to_f<T>(pb[e]) * drop_mult(dc, (unsigned long)(row * C + c0 + e));
        s += v[pss][e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < V; ++e) v[pss][e] = 0.f;
    }
  }
  const float mu = wave_reduce_sum(s) / C;
  float q = 0.f;
#pragma unroll
  for (int pss = 0; pss < NP; ++pss) {
    const int c0 = (pss * 64 + lane) * V;
    if (c0 < C)
#pragma unroll
      for (int e = 0; e < V; ++e) { const float d = v[pss][e] - mu; q += d * d; }
  }
  const float rs = rsqrtf(wave_reduce_sum(q) / C + eps);
#pragma unroll
  for (int pss = 0; pss < NP; ++pss) {
    const int c0 = (pss * 64 + lane) * V;
    if (c0 < C) {
      uint4 o;
      T* po = reinterpret_cast<T*>(&o);
#pragma unroll
      for (int e = 0; e < V; ++e) po[e] = from_f<T>((v[pss][e] - mu) * rs * gamma[c0 + e] + beta[c0 + e]);
      *reinterpret_cast<uint4*>(out + row * C + c0) = o;
    }
  }
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

// dx = d(x + drop(y)) (residual branch), dy = dx * dropout multiplier (sub-layer branch; null when p == 0: same tensor)
// WAVES: waves per block.  The pass is a stream (3 reads + 1-2 writes of [rows, C]); with 256 blocks of 4 waves a CU held
// one wave per SIMD and 12 KB of loads in flight: 2.6 TB/s on the s1 shapes ([32768, 512]: 52 us).  Long inputs take 8 waves
// per block and up to 512 blocks (4 waves per SIMD); the per-channel atomics double to 0.5 M, still one per block and channel.
template <typename T, int NP, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void res_drop_ln_bwd(const T* x, const T* y, const float* gamma, const T* dout,
                                                       const float* mean, const float* rstd, const int* lens,
                                                       int rows_per_seq, float p, const unsigned* seed_dev,
                                                       unsigned site, T* dx, T* dy, float* dgamma, float* dbeta,
                                                       long rows, int C, int rows_per_block) {
  constexpr int V = Vec16<T>::V;
  __shared__ float sg[WAVES][64 * V * NP], sb[WAVES][64 * V * NP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const DropCfg dc = drop_cfg(p, seed_dev, site);
  float ag[NP][V], ab[NP][V], gm[NP][V];
#pragma unroll
  for (int pss = 0; pss < NP; ++pss)
#pragma unroll
    for (int e = 0; e < V; ++e) {
      ag[pss][e] = ab[pss][e] = 0.f;
      const int c = (pss * 64 + lane) * V + e;
      gm[pss][e] = c < C ? gamma[c] : 0.f;
    }
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = min(rows, r0 + rows_per_block);
  for (long row = r0 + wave; row < r1; row += WAVES) {
    bool live = true;
    if (lens) { const long b = row / rows_per_seq; live = (int)(row - b * rows_per_seq) < lens[b]; }
    if (!live) {
#pragma unroll
      for (int pss = 0; pss < NP; ++pss) {
        const int c0 = (pss * 64 + lane) * V;
        if (c0 < C) {
          *reinterpret_cast<uint4*>(dx + row * C + c0) = make_uint4(0, 0, 0, 0);
          if (dy) *reinterpret_cast<uint4*>(dy + row * C + c0) = make_uint4(0, 0, 0, 0);
        }
      }
      continue;
    }
    const float mu = mean[row], rs = rstd[row];
    float xh[NP][V], dh[NP][V], dm[NP][V];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int pss = 0; pss < NP; ++pss) {
      const int c0 = (pss * 64 + lane) * V;
      if (c0 < C) {
        const uint4 a = *reinterpret_cast<const uint4*>(x + row * C + c0);
        const uint4 b = *reinterpret_cast<const uint4*>(y + row * C + c0);
        const uint4 d4 = *reinterpret_cast<const uint4*>(dout + row * C + c0);
        const T* pa = reinterpret_cast<const T*>(&a);
        const T* pb = reinterpret_cast<const T*>(&b);
        const T* pd = reinterpret_cast<const T*>(&d4);
#pragma unroll
        for (int e = 0; e < V; ++e) {
          dm[pss][e] = drop_mult(dc, (unsigned long)(row * C + c0 + e));
          const float t = to_f<T>(pa[e]) + to_f<T>(pb[e]) * dm[pss][e];
          const float d = to_f<T>(pd[e]);
          xh[pss][e] = (t - mu) * rs;
          dh[pss][e] = d * gm[pss][e];
          ag[pss][e] += d * xh[pss][e];
          ab[pss][e] += d;
          s1 += dh[pss][e];
          s2 += dh[pss][e] * xh[pss][e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < V; ++e) xh[pss][e] = dh[pss][e] = dm[pss][e] = 0.f;
      }
    }
    s1 = wave_reduce_sum(s1) / C;
    s2 = wave_reduce_sum(s2) / C;
#pragma unroll
    for (int pss = 0; pss < NP; ++pss) {
      const int c0 = (pss * 64 + lane) * V;
      if (c0 < C) {
        uint4 o, o2;
        T* po = reinterpret_cast<T*>(&o);
        T* po2 = reinterpret_cast<T*>(&o2);
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const float g = rs * (dh[pss][e] - s1 - xh[pss][e] * s2);
          po[e] = from_f<T>(g);
          po2[e] = from_f<T>(g * dm[pss][e]);
        }
        *reinterpret_cast<uint4*>(dx + row * C + c0) = o;
        if (dy) *reinterpret_cast<uint4*>(dy + row * C + c0) = o2;
      }
    }
  }
#pragma unroll
  for (int pss = 0; pss < NP; ++pss)
#pragma unroll
    for (int e = 0; e < V; ++e) {
      sg[wave][(pss * 64 + lane) * V + e] = ag[pss][e];
      sb[wave][(pss * 64 + lane) * V + e] = ab[pss][e];
    }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 64 * WAVES) {
    float tg = 0.f, tb = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) { tg += sg[w][c]; tb += sb[w][c]; }
    atomicAdd(dgamma + c, tg);
    atomicAdd(dbeta + c, tb);
  }
}


// ---- WN layer glue (modules.py:199-211 of the reference): res_skip output rs [rows][2H] ->
//        x_new = (x + rs[:, :H]) * row_mask          acc_new = acc + rs[:, H:]
//      last layer (rs has H channels):               acc_new = (acc + rs) * row_mask
//      one launch instead of slice / add / mul / add (and their ~8 backward launches) ----
template <typename T>
__global__ void wn_residual_fwd(const T* x, const T* rs, const T* acc, const int* lens, int rows_per_seq, T* x_out,
                                T* acc_out, long rows, int H, int last) {
  constexpr int V = 16 / sizeof(T);
  const int ppr = H / V;
  const long total = rows * ppr;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / ppr;
    const int c0 = (int)(i - row * ppr) * V;
    bool live = true;
    if (lens) { const long b = row / rows_per_seq; live = (int)(row - b * rows_per_seq) < lens[b]; }
    const int ldr = last ? H : 2 * H;
    const uint4 r0 = *reinterpret_cast<const uint4*>(rs + row * ldr + c0);
    uint4 av = make_uint4(0, 0, 0, 0);
    if (acc) av = *reinterpret_cast<const uint4*>(acc + row * H + c0);
    const T* pr0 = reinterpret_cast<const T*>(&r0);
    const T* pa = reinterpret_cast<const T*>(&av);
    uint4 o1, o2;
    T* p1 = reinterpret_cast<T*>(&o1);
    T* p2 = reinterpret_cast<T*>(&o2);
    if (last) {
#pragma unroll
      for (int e = 0; e < V; ++e) p2[e] = live ? from_f<T>(to_f<T>(pa[e]) + to_f<T>(pr0[e])) : from_f<T>(0.f);
      *reinterpret_cast<uint4*>(acc_out + row * H + c0) = o2;
    } else {
      const uint4 xv = *reinterpret_cast<const uint4*>(x + row * H + c0);
      const uint4 r1 = *reinterpret_cast<const uint4*>(rs + row * ldr + H + c0);
      const T* px = reinterpret_cast<const T*>(&xv);
      const T* pr1 = reinterpret_cast<const T*>(&r1);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        p1[e] = live ? from_f<T>(to_f<T>(px[e]) + to_f<T>(pr0[e])) : from_f<T>(0.f);
        p2[e] = from_f<T>(to_f<T>(pa[e]) + to_f<T>(pr1[e]));
      }
      *reinterpret_cast<uint4*>(x_out + row * H + c0) = o1;
      *reinterpret_cast<uint4*>(acc_out + row * H + c0) = o2;
    }
  }
}

// drs[:, :H] = dx_out * mask (also written to dx), drs[:, H:] = dacc_out;   last: drs = dacc_out * mask
template <typename T>
__global__ void wn_residual_bwd(const T* dx_out, const T* dacc_out, const int* lens, int rows_per_seq, T* dx, T* drs,
                                long rows, int H, int last) {
  constexpr int V = 16 / sizeof(T);
  const int ppr = H / V;
  const long total = rows * ppr;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / ppr;
    const int c0 = (int)(i - row * ppr) * V;
    bool live = true;
    if (lens) { const long b = row / rows_per_seq; live = (int)(row - b * rows_per_seq) < lens[b]; }
    const uint4 z = make_uint4(0, 0, 0, 0);
    uint4 da = z;
    if (dacc_out) da = *reinterpret_cast<const uint4*>(dacc_out + row * H + c0);
    if (last) {
      *reinterpret_cast<uint4*>(drs + row * H + c0) = live ? da : z;
    } else {
      uint4 dxv = z;
      if (dx_out && live) dxv = *reinterpret_cast<const uint4*>(dx_out + row * H + c0);
      *reinterpret_cast<uint4*>(dx + row * H + c0) = dxv;
      *reinterpret_cast<uint4*>(drs + row * 2 * H + c0) = dxv;
      *reinterpret_cast<uint4*>(drs + row * 2 * H + H + c0) = da;
    }
  }
}

// ---- y = dropout(relu(x)) and its backward dx = dy * mult * (x > 0): the FFN inner activation of the s1 blocks
//      (transformer.py:330-334 of the reference: linear2(dropout(relu(linear1(x))))), one pass each way ----
// optional row mask: element index -> row = idx / C, live iff (row % rows_per_seq) < lens[row / rows_per_seq]
__device__ __forceinline__ bool row_live(const int* lens, int rows_per_seq, int C, long elem) {
  if (!lens) return true;
  const long row = elem / C;
  const long b = row / rows_per_seq;
  return (int)(row - b * rows_per_seq) < lens[b];
}

template <typename T>
__global__ void relu_dropout_fwd(const T* x, float p, const unsigned* seed_dev, unsigned site, const int* lens,
                                 int rows_per_seq, int C, T* y, long n) {
  constexpr int V = 16 / sizeof(T);
  const DropCfg dc = drop_cfg(p, seed_dev, site);
  const long nv = n / V;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
    uint4 v = reinterpret_cast<const uint4*>(x)[i];
    T* h = reinterpret_cast<T*>(&v);
    const bool live = row_live(lens, rows_per_seq, C, i * V);     // C % V == 0: a piece never straddles rows
#pragma unroll
    for (int e = 0; e < V; ++e)
      h[e] = from_f<T>(live ? fmaxf(to_f<T>(h[e]), 0.f) * drop_mult(dc, (unsigned long)(i * V + e)) : 0.f);
    reinterpret_cast<uint4*>(y)[i] = v;
  }
}

template <typename T>
__global__ void relu_dropout_bwd(const T* x, const T* dy, float p, const unsigned* seed_dev, unsigned site, const int* lens,
                                 int rows_per_seq, int C, T* dx, long n) {
  constexpr int V = 16 / sizeof(T);
  const DropCfg dc = drop_cfg(p, seed_dev, site);
  const long nv = n / V;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
    const uint4 xv = reinterpret_cast<const uint4*>(x)[i];
    uint4 dv = reinterpret_cast<const uint4*>(dy)[i];
    const T* px = reinterpret_cast<const T*>(&xv);
    T* pd = reinterpret_cast<T*>(&dv);
    const bool live = row_live(lens, rows_per_seq, C, i * V);
#pragma unroll
    for (int e = 0; e < V; ++e)
      pd[e] = from_f<T>(live && to_f<T>(px[e]) > 0.f ? to_f<T>(pd[e]) * drop_mult(dc, (unsigned long)(i * V + e)) : 0.f);
    reinterpret_cast<uint4*>(dx)[i] = dv;
  }
}

__global__ void counter_add_kernel(unsigned* c, unsigned inc) { *c += inc; }


// ---- style encoder element-wise chains (src/easevoice/module/modules.py:521-566, 685-763) ----
// Mish + dropout:  y = drop(x * tanh(softplus(x)))  -- torch: cast, softplus, tanh, mul, dropout (5 launches, 6 backward)
__device__ __forceinline__ float mish_f(float x) {
  const float sp = x > 20.f ? x : log1pf(__expf(x));       // F.softplus: threshold 20
  return x * tanhf(sp);
}
__device__ __forceinline__ float mish_grad(float x) {
  const float sp = x > 20.f ? x : log1pf(__expf(x));
  const float t = tanhf(sp);
  const float sg = 1.f / (1.f + __expf(-x));                // d softplus / dx
  return t + x * (1.f - t * t) * sg;
}
// x in T, y (and dy) in TY: the reference's autocast leaves Mish / the GLU residual stream in fp32 next to half-precision
// projections; [B, T, 128] tensors -- launch-bound, one element per thread and pass
template <typename T, typename TY>
__global__ void mish_dropout_fwd(const T* x, float p, const unsigned* seed_dev, unsigned site, TY* y, long n) {
  const DropCfg dc = drop_cfg(p, seed_dev, site);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = from_f<TY>(mish_f(to_f<T>(x[i])) * drop_mult(dc, (unsigned long)i));
}
template <typename T, typename TY>
__global__ void mish_dropout_bwd(const T* x, const TY* dy, float p, const unsigned* seed_dev, unsigned site, T* dx, long n) {
  const DropCfg dc = drop_cfg(p, seed_dev, site);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dx[i] = from_f<T>(to_f<TY>(dy[i]) * drop_mult(dc, (unsigned long)i) * mish_grad(to_f<T>(x[i])));
}
// Conv1dGLU tail:  y = res + drop(h[:, :C] * sigmoid(h[:, C:]))  -- torch: sigmoid, mul, dropout, add (+ 6 backward);
// h (and dh) in T, res / y / dy in TR
template <typename T, typename TR>
__global__ void glu_dropout_res_fwd(const T* h, const TR* res, float p, const unsigned* seed_dev, unsigned site, TR* y,
                                    long rows, int C) {
  const DropCfg dc = drop_cfg(p, seed_dev, site);
  const long n = rows * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long row = i / C;
    const int c = (int)(i - row * C);
    const float sg = 1.f / (1.f + __expf(-to_f<T>(h[row * 2 * C + C + c])));
    y[i] = from_f<TR>(to_f<TR>(res[i]) + to_f<T>(h[row * 2 * C + c]) * sg * drop_mult(dc, (unsigned long)i));
  }
}
template <typename T, typename TR>
__global__ void glu_dropout_res_bwd(const T* h, const TR* dy, float p, const unsigned* seed_dev, unsigned site, T* dh,
                                    long rows, int C) {
  const DropCfg dc = drop_cfg(p, seed_dev, site);
  const long n = rows * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long row = i / C;
    const int c = (int)(i - row * C);
    const float a = to_f<T>(h[row * 2 * C + c]);
    const float sg = 1.f / (1.f + __expf(-to_f<T>(h[row * 2 * C + C + c])));
    const float g = to_f<TR>(dy[i]) * drop_mult(dc, (unsigned long)i);
    dh[row * 2 * C + c] = from_f<T>(g * sg);
    dh[row * 2 * C + C + c] = from_f<T>(g * a * sg * (1.f - sg));
  }
}

// ---- posterior encoder tail (src/easevoice/module/models.py:352-358):  stats = proj(h) * mask;  m, logs = split(stats);
//      z = (m + eps * exp(logs)) * mask  -- mask, cast, split, exp, multiply, add, mask as one launch (and one backward) ----
template <typename T>
__global__ void reparam_fwd(const T* stats, const float* eps, const int* lens, int rows_per_seq, long rows, int C, float* z,
                            float* m, float* logs) {
  const long total = rows * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / C;
    const int c = (int)(i - row * C);
    bool live = true;
    if (lens) { const long b = row / rows_per_seq; live = (int)(row - b * rows_per_seq) < lens[b]; }
    const float mv = live ? to_f<T>(stats[row * 2 * C + c]) : 0.f;
    const float lv = live ? to_f<T>(stats[row * 2 * C + C + c]) : 0.f;
    m[i] = mv;
    logs[i] = lv;
    z[i] = live ? mv + eps[i] * __expf(lv) : 0.f;
  }
}
template <typename T>
__global__ void reparam_bwd(const float* dz, const float* dm, const float* dlogs, const float* eps, const float* logs,
                            const int* lens, int rows_per_seq, long rows, int C, T* dstats) {
  const long total = rows * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / C;
    const int c = (int)(i - row * C);
    bool live = true;
    if (lens) { const long b = row / rows_per_seq; live = (int)(row - b * rows_per_seq) < lens[b]; }
    float gm = 0.f, gl = 0.f;
    if (live) {
      const float gz = dz ? dz[i] : 0.f;
      gm = gz + (dm ? dm[i] : 0.f);
      gl = gz * eps[i] * __expf(logs[i]) + (dlogs ? dlogs[i] : 0.f);
    }
    dstats[row * 2 * C + c] = from_f<T>(gm);
    dstats[row * 2 * C + C + c] = from_f<T>(gl);
  }
}

// ---- mean-only residual coupling + Flip of the s2 flow, everything after the layer's `post` projection as one launch:
//        y = flip_channels( [ x0 , (x1 + stats) * row_mask ] ),   x0n = first half of y in the compute dtype (what the next
//      layer's `pre` projection reads).  Through torch: mask multiply, cast, multiply, add, cat, flip and the slice + cast of
//      the next layer's input (7 launches, ~12 backward); the tensors are [B, T, 192] -- launch-bound ----
template <typename T>
__global__ void coupling_flip_fwd(const float* x, const T* stats, const int* lens, int rows_per_seq, long rows, int h,
                                  float* y, T* x0n) {
  const int C2 = 2 * h;
  const long total = rows * C2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / C2;
    const int c = (int)(i - row * C2);
    const int cs = C2 - 1 - c;
    float v = x[row * C2 + cs];
    if (cs >= h) {
      bool live = true;
      if (lens) { const long b = row / rows_per_seq; live = (int)(row - b * rows_per_seq) < lens[b]; }
      v = live ? v + to_f<T>(stats[row * h + cs - h]) : 0.f;
    }
    y[i] = v;
    if (x0n && c < h) x0n[row * h + c] = from_f<T>(v);
  }
}

// dy [rows][2h] (+ dx0n [rows][h] on its first half) -> dx [rows][2h], dstats [rows][h]
template <typename T>
__global__ void coupling_flip_bwd(const float* dy, const T* dx0n, const int* lens, int rows_per_seq, long rows, int h,
                                  float* dx, T* dstats) {
  const int C2 = 2 * h;
  const long total = rows * C2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / C2;
    const int c = (int)(i - row * C2);
    const int cs = C2 - 1 - c;
    float t = dy[i];
    if (dx0n && c < h) t += to_f<T>(dx0n[row * h + c]);
    if (cs >= h) {
      bool live = true;
      if (lens) { const long b = row / rows_per_seq; live = (int)(row - b * rows_per_seq) < lens[b]; }
      t = live ? t : 0.f;
      dstats[row * h + cs - h] = from_f<T>(t);
    }
    dx[row * C2 + cs] = t;
  }
}

}  // namespace

static inline dim3 ew_grid(long nv) {
  long blocks = (nv + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  return dim3((int)blocks);
}

// dtype: x / h (and dx / dh); wide_dtype: y / res / dy -- the same, or fp32 next to bf16 operands.  f(T*, TR*) launches
// the instantiation for the pair.
template <typename F>
static int mixed_dispatch(int32_t dtype, int32_t wide_dtype, F&& f) {
  if (dtype == EVT_DT_HALF && wide_dtype == EVT_DT_HALF) f((h16_t*)nullptr, (h16_t*)nullptr);
  else if (dtype == EVT_DT_HALF && wide_dtype == EVT_DT_F32) f((h16_t*)nullptr, (float*)nullptr);
  else if (dtype == EVT_DT_F32 && wide_dtype == EVT_DT_F32) f((float*)nullptr, (float*)nullptr);
  else return EVT_EINVAL;
  return evt_check_launch();
}

extern "C" {

int evt_counter_inc(uint32_t* counter, uint32_t inc, void* stream) {
  if (!counter) return EVT_EINVAL;
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter, inc);
  return evt_check_launch();
}

int evt_res_dropout_ln_fwd(int32_t dtype, const void* x, const void* y, const float* gamma, const float* beta,
                           const int32_t* lens, int32_t rows_per_seq, float p, const uint32_t* seed_dev, uint32_t site,
                           void* out, float* mean, float* rstd, int64_t rows, int32_t C, float eps, void* stream) {
  if (!x || !y || !gamma || !beta || !out || !mean || !rstd || rows <= 0 || C <= 0) return EVT_EINVAL;
  if (lens && rows_per_seq <= 0) return EVT_EINVAL;
  if (p < 0.f || p >= 1.f) return EVT_EINVAL;
  if (C > 1024 || C % 8) return EVT_ENOTSUP;
  hipStream_t st = (hipStream_t)stream;
  const int blocks = (int)((rows + 3) / 4);
#define RDL_FWD(T, E) hipLaunchKernelGGL((res_drop_ln_fwd<T, E>), dim3(blocks), dim3(256), 0, st, (const T*)x, (const T*)y, \
                                         gamma, beta, lens, rows_per_seq, p, seed_dev, site, (T*)out, mean, rstd, (long)rows, C, eps)
  if (dtype == EVT_DT_HALF) { if (C <= 512) RDL_FWD(h16_t, 1); else RDL_FWD(h16_t, 2); }
  else if (dtype == EVT_DT_F32) { if (C <= 512) RDL_FWD(float, 2); else RDL_FWD(float, 4); }
  else return EVT_EINVAL;
#undef RDL_FWD
  return evt_check_launch();
}

int evt_res_dropout_ln_bwd(int32_t dtype, const void* x, const void* y, const float* gamma, const void* dout,
                           const float* mean, const float* rstd, const int32_t* lens, int32_t rows_per_seq, float p,
                           const uint32_t* seed_dev, uint32_t site, void* dx, void* dy, float* dgamma, float* dbeta,
                           int64_t rows, int32_t C, void* stream) {
  if (!x || !y || !gamma || !dout || !mean || !rstd || !dx || !dgamma || !dbeta || rows <= 0 || C <= 0) return EVT_EINVAL;
  if (lens && rows_per_seq <= 0) return EVT_EINVAL;
  if (p < 0.f || p >= 1.f) return EVT_EINVAL;
  if (p > 0.f && !dy) return EVT_EINVAL;
  if (C > 1024 || C % 8) return EVT_ENOTSUP;
  hipStream_t st = (hipStream_t)stream;
  static const int wide_off = getenv("EVT_LN_BWD_WAVES4") != nullptr;      // A/B switch for measurements
  const bool wide = rows >= 8192 && !wide_off;
  const int maxb = wide ? 512 : 256;            // each block ends with one atomic per channel
  long rpb = (rows + maxb - 1) / maxb;
  if (rpb < 16) rpb = 16;
  const int blocks = (int)((rows + rpb - 1) / rpb);
#define RDL_BWD(T, E, W) hipLaunchKernelGGL((res_drop_ln_bwd<T, E, W>), dim3(blocks), dim3(64 * W), 0, st, (const T*)x,      \
                                            (const T*)y, gamma, (const T*)dout, mean, rstd, lens, rows_per_seq, p, seed_dev, \
                                            site, (T*)dx, (T*)dy, dgamma, dbeta, (long)rows, C, (int)rpb)
  if (dtype == EVT_DT_HALF) {
    if (C <= 512) { if (wide) RDL_BWD(h16_t, 1, 8); else RDL_BWD(h16_t, 1, 4); }
    else { if (wide) RDL_BWD(h16_t, 2, 8); else RDL_BWD(h16_t, 2, 4); }
  } else if (dtype == EVT_DT_F32) {
    if (C <= 512) RDL_BWD(float, 2, 4); else RDL_BWD(float, 4, 4);
  } else return EVT_EINVAL;
#undef RDL_BWD
  return evt_check_launch();
}


int evt_reparam_fwd(int32_t dtype, const void* stats, const float* eps, const int32_t* lens, int32_t rows_per_seq,
                    int64_t rows, int32_t C, float* z, float* m, float* logs, void* stream) {
  if (!stats || !eps || !z || !m || !logs || rows <= 0 || C <= 0 || (lens && rows_per_seq <= 0)) return EVT_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EVT_DT_HALF)
    hipLaunchKernelGGL(reparam_fwd<h16_t>, ew_grid(rows * C), dim3(256), 0, st, (const h16_t*)stats, eps, lens, rows_per_seq,
                       (long)rows, C, z, m, logs);
  else if (dtype == EVT_DT_F32)
    hipLaunchKernelGGL(reparam_fwd<float>, ew_grid(rows * C), dim3(256), 0, st, (const float*)stats, eps, lens, rows_per_seq,
                       (long)rows, C, z, m, logs);
  else return EVT_EINVAL;
  return evt_check_launch();
}

int evt_reparam_bwd(int32_t dtype, const float* dz, const float* dm, const float* dlogs, const float* eps, const float* logs,
                    const int32_t* lens, int32_t rows_per_seq, int64_t rows, int32_t C, void* dstats, void* stream) {
  if (!eps || !logs || !dstats || rows <= 0 || C <= 0 || (lens && rows_per_seq <= 0)) return EVT_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EVT_DT_HALF)
    hipLaunchKernelGGL(reparam_bwd<h16_t>, ew_grid(rows * C), dim3(256), 0, st, dz, dm, dlogs, eps, logs, lens, rows_per_seq,
                       (long)rows, C, (h16_t*)dstats);
  else if (dtype == EVT_DT_F32)
    hipLaunchKernelGGL(reparam_bwd<float>, ew_grid(rows * C), dim3(256), 0, st, dz, dm, dlogs, eps, logs, lens, rows_per_seq,
                       (long)rows, C, (float*)dstats);
  else return EVT_EINVAL;
  return evt_check_launch();
}

#define EVT_TT(t, tr) using T = std::remove_pointer_t<decltype(t)>; using TR = std::remove_pointer_t<decltype(tr)>

int evt_mish_dropout_fwd(int32_t dtype, int32_t wide_dtype, const void* x, float p, const uint32_t* seed_dev, uint32_t site,
                         void* y, int64_t n, void* stream) {
  if (!x || !y || n <= 0 || p < 0.f || p >= 1.f) return EVT_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  return mixed_dispatch(dtype, wide_dtype, [&](auto* t, auto* tr) {
    EVT_TT(t, tr);
    hipLaunchKernelGGL((mish_dropout_fwd<T, TR>), ew_grid(n), dim3(256), 0, st, (const T*)x, p, seed_dev, site, (TR*)y, (long)n);
  });
}

int evt_mish_dropout_bwd(int32_t dtype, int32_t wide_dtype, const void* x, const void* dy, float p, const uint32_t* seed_dev,
                         uint32_t site, void* dx, int64_t n, void* stream) {
  if (!x || !dy || !dx || n <= 0 || p < 0.f || p >= 1.f) return EVT_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  return mixed_dispatch(dtype, wide_dtype, [&](auto* t, auto* tr) {
    EVT_TT(t, tr);
    hipLaunchKernelGGL((mish_dropout_bwd<T, TR>), ew_grid(n), dim3(256), 0, st, (const T*)x, (const TR*)dy, p, seed_dev, site,
                       (T*)dx, (long)n);
  });
}

int evt_glu_dropout_res_fwd(int32_t dtype, int32_t wide_dtype, const void* h, const void* res, float p,
                            const uint32_t* seed_dev, uint32_t site, void* y, int64_t rows, int32_t C, void* stream) {
  if (!h || !res || !y || rows <= 0 || C <= 0 || p < 0.f || p >= 1.f) return EVT_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  return mixed_dispatch(dtype, wide_dtype, [&](auto* t, auto* tr) {
    EVT_TT(t, tr);
    hipLaunchKernelGGL((glu_dropout_res_fwd<T, TR>), ew_grid(rows * C), dim3(256), 0, st, (const T*)h, (const TR*)res, p,
                       seed_dev, site, (TR*)y, (long)rows, C);
  });
}

int evt_glu_dropout_res_bwd(int32_t dtype, int32_t wide_dtype, const void* h, const void* dy, float p,
                            const uint32_t* seed_dev, uint32_t site, void* dh, int64_t rows, int32_t C, void* stream) {
  if (!h || !dy || !dh || rows <= 0 || C <= 0 || p < 0.f || p >= 1.f) return EVT_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  return mixed_dispatch(dtype, wide_dtype, [&](auto* t, auto* tr) {
    EVT_TT(t, tr);
    hipLaunchKernelGGL((glu_dropout_res_bwd<T, TR>), ew_grid(rows * C), dim3(256), 0, st, (const T*)h, (const TR*)dy, p,
                       seed_dev, site, (T*)dh, (long)rows, C);
  });
}

int evt_coupling_flip_fwd(int32_t dtype, const float* x, const void* stats, const int32_t* lens, int32_t rows_per_seq,
                          int64_t rows, int32_t h, float* y, void* x0n, void* stream) {
  if (!x || !stats || !y || rows <= 0 || h <= 0 || (lens && rows_per_seq <= 0)) return EVT_EINVAL;
  long blocks = (rows * 2 * h + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EVT_DT_HALF)
    hipLaunchKernelGGL(coupling_flip_fwd<h16_t>, dim3((int)blocks), dim3(256), 0, st, x, (const h16_t*)stats, lens,
                       rows_per_seq, (long)rows, h, y, (h16_t*)x0n);
  else if (dtype == EVT_DT_F32)
    hipLaunchKernelGGL(coupling_flip_fwd<float>, dim3((int)blocks), dim3(256), 0, st, x, (const float*)stats, lens,
                       rows_per_seq, (long)rows, h, y, (float*)x0n);
  else return EVT_EINVAL;
  return evt_check_launch();
}

int evt_coupling_flip_bwd(int32_t dtype, const float* dy, const void* dx0n, const int32_t* lens, int32_t rows_per_seq,
                          int64_t rows, int32_t h, float* dx, void* dstats, void* stream) {
  if (!dy || !dx || !dstats || rows <= 0 || h <= 0 || (lens && rows_per_seq <= 0)) return EVT_EINVAL;
  long blocks = (rows * 2 * h + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EVT_DT_HALF)
    hipLaunchKernelGGL(coupling_flip_bwd<h16_t>, dim3((int)blocks), dim3(256), 0, st, dy, (const h16_t*)dx0n, lens,
                       rows_per_seq, (long)rows, h, dx, (h16_t*)dstats);
  else if (dtype == EVT_DT_F32)
    hipLaunchKernelGGL(coupling_flip_bwd<float>, dim3((int)blocks), dim3(256), 0, st, dy, (const float*)dx0n, lens,
                       rows_per_seq, (long)rows, h, dx, (float*)dstats);
  else return EVT_EINVAL;
  return evt_check_launch();
}

int evt_wn_residual_fwd(int32_t dtype, const void* x, const void* rs, const void* acc, const int32_t* lens,
                        int32_t rows_per_seq, void* x_out, void* acc_out, int64_t rows, int32_t H, int32_t last,
                        void* stream) {
  if (!rs || !acc_out || rows <= 0 || H <= 0 || (!last && (!x || !x_out))) return EVT_EINVAL;
  if (lens && rows_per_seq <= 0) return EVT_EINVAL;
  const int V = dtype == EVT_DT_HALF ? 8 : 4;
  if (H % V) return EVT_ENOTSUP;
  const long total = rows * (H / V);
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EVT_DT_HALF)
    hipLaunchKernelGGL(wn_residual_fwd<h16_t>, dim3((int)blocks), dim3(256), 0, st, (const h16_t*)x, (const h16_t*)rs,
                       (const h16_t*)acc, lens, rows_per_seq, (h16_t*)x_out, (h16_t*)acc_out, (long)rows, H, last);
  else if (dtype == EVT_DT_F32)
    hipLaunchKernelGGL(wn_residual_fwd<float>, dim3((int)blocks), dim3(256), 0, st, (const float*)x, (const float*)rs,
                       (const float*)acc, lens, rows_per_seq, (float*)x_out, (float*)acc_out, (long)rows, H, last);
  else return EVT_EINVAL;
  return evt_check_launch();
}

int evt_wn_residual_bwd(int32_t dtype, const void* dx_out, const void* dacc_out, const int32_t* lens,
                        int32_t rows_per_seq, void* dx, void* drs, int64_t rows, int32_t H, int32_t last, void* stream) {
  if (!drs || rows <= 0 || H <= 0 || (!last && !dx)) return EVT_EINVAL;
  if (lens && rows_per_seq <= 0) return EVT_EINVAL;
  const int V = dtype == EVT_DT_HALF ? 8 : 4;
  if (H % V) return EVT_ENOTSUP;
  const long total = rows * (H / V);
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EVT_DT_HALF)
    hipLaunchKernelGGL(wn_residual_bwd<h16_t>, dim3((int)blocks), dim3(256), 0, st, (const h16_t*)dx_out,
                       (const h16_t*)dacc_out, lens, rows_per_seq, (h16_t*)dx, (h16_t*)drs, (long)rows, H, last);
  else if (dtype == EVT_DT_F32)
    hipLaunchKernelGGL(wn_residual_bwd<float>, dim3((int)blocks), dim3(256), 0, st, (const float*)dx_out,
                       (const float*)dacc_out, lens, rows_per_seq, (float*)dx, (float*)drs, (long)rows, H, last);
  else return EVT_EINVAL;
  return evt_check_launch();
}

int evt_relu_dropout_fwd(int32_t dtype, const void* x, float p, const uint32_t* seed_dev, uint32_t site,
                         const int32_t* lens, int32_t rows_per_seq, int32_t C, void* y, int64_t n, void* stream) {
  if (!x || !y || n <= 0 || p < 0.f || p >= 1.f) return EVT_EINVAL;
  const int V = dtype == EVT_DT_HALF ? 8 : 4;
  if (lens && (rows_per_seq <= 0 || C <= 0 || C % V)) return EVT_EINVAL;
  if (n % V || (((uintptr_t)x | (uintptr_t)y) & 15)) return EVT_EINVAL;
  long blocks = (n / V + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EVT_DT_HALF)
    hipLaunchKernelGGL(relu_dropout_fwd<h16_t>, dim3((int)blocks), dim3(256), 0, st, (const h16_t*)x, p, seed_dev, site,
                       lens, rows_per_seq, C, (h16_t*)y, (long)n);
  else if (dtype == EVT_DT_F32)
    hipLaunchKernelGGL(relu_dropout_fwd<float>, dim3((int)blocks), dim3(256), 0, st, (const float*)x, p, seed_dev, site,
                       lens, rows_per_seq, C, (float*)y, (long)n);
  else return EVT_EINVAL;
  return evt_check_launch();
}

int evt_relu_dropout_bwd(int32_t dtype, const void* x, const void* dy, float p, const uint32_t* seed_dev, uint32_t site,
                         const int32_t* lens, int32_t rows_per_seq, int32_t C, void* dx, int64_t n, void* stream) {
  if (!x || !dy || !dx || n <= 0 || p < 0.f || p >= 1.f) return EVT_EINVAL;
  const int V = dtype == EVT_DT_HALF ? 8 : 4;
  if (lens && (rows_per_seq <= 0 || C <= 0 || C % V)) return EVT_EINVAL;
  if (n % V || (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15)) return EVT_EINVAL;
  long blocks = (n / V + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EVT_DT_HALF)
    hipLaunchKernelGGL(relu_dropout_bwd<h16_t>, dim3((int)blocks), dim3(256), 0, st, (const h16_t*)x, (const h16_t*)dy,
                       p, seed_dev, site, lens, rows_per_seq, C, (h16_t*)dx, (long)n);
  else if (dtype == EVT_DT_F32)
    hipLaunchKernelGGL(relu_dropout_bwd<float>, dim3((int)blocks), dim3(256), 0, st, (const float*)x, (const float*)dy, p,
                       seed_dev, site, lens, rows_per_seq, C, (float*)dx, (long)n);
  else return EVT_EINVAL;
  return evt_check_launch();
}

}  // extern "C"
