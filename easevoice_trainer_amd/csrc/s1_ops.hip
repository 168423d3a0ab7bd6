// s1 (text->semantic GPT) support kernels for gfx950: fused residual-add + LayerNorm, cross-entropy(sum) with the
// gradient and the top-k hit count in the same pass, and ScaledAdam over a flat parameter arena.
//
// Reference call sites (file:line under /root/reference):
//   post-LN block        src/easevoice/soundstorm/auto_reg/modules/transformer.py:311-315 (norm(x + sublayer(x)))
//   CE(sum) + top-3 acc  src/easevoice/soundstorm/auto_reg/models/t2s_model.py:486-489, :301-307
//   ScaledAdam           src/easevoice/soundstorm/auto_reg/modules/optim.py:206-251,300-390,448-622
#include "evt_common.h"
#include "../../include/evt.h"

namespace {

// ---- y = LayerNorm(x + r): one wave per row, 16-byte loads; a lane owns V = 16/sizeof(T) consecutive channels per
//      pass (C % V == 0, C <= 64*V*NP) -------------------------------------------------------------------------------
template <typename T> struct Vec16 { static constexpr int V = 16 / sizeof(T); };

template <typename T, int NP>
__global__ __launch_bounds__(256) void add_ln_fwd(const T* x, const T* r, const float* gamma, const float* beta, T* y,
                                                  float* mean, float* rstd, long rows, int C, float eps) {
  constexpr int V = Vec16<T>::V;
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float v[NP][V];
  float s = 0.f;
#pragma unroll
  for (int pss = 0; pss < NP; ++pss) {
    const int c0 = (pss * 64 + lane) * V;
    if (c0 < C) {
      const uint4 a = *reinterpret_cast<const uint4*>(x + row * C + c0);
      uint4 b = make_uint4(0, 0, 0, 0);
      if (r) b = *reinterpret_cast<const uint4*>(r + row * C + c0);
      const T* pa = reinterpret_cast<const T*>(&a);
      const T* pb = reinterpret_cast<const T*>(&b);
#pragma unroll
      for (int e = 0; e < V; ++e) { v[pss][e] = to_f<T>(pa[e]) + (r ? to_f<T>(pb[e]) : 0.f); s += v[pss][e]; }
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
      *reinterpret_cast<uint4*>(y + row * C + c0) = o;
    }
  }
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

template <typename T, int NP>
__global__ __launch_bounds__(256) void add_ln_bwd(const T* x, const T* r, const float* gamma, const T* dy,
                                                  const float* mean, const float* rstd, T* dxr, float* dgamma,
                                                  float* dbeta, long rows, int C, int rows_per_block) {
  constexpr int V = Vec16<T>::V;
  __shared__ float sg[4][64 * V * NP], sb[4][64 * V * NP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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
  for (long row = r0 + wave; row < r1; row += 4) {
    const float mu = mean[row], rs = rstd[row];
    float xh[NP][V], dh[NP][V];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int pss = 0; pss < NP; ++pss) {
      const int c0 = (pss * 64 + lane) * V;
      if (c0 < C) {
        const uint4 a = *reinterpret_cast<const uint4*>(x + row * C + c0);
        uint4 b = make_uint4(0, 0, 0, 0);
        if (r) b = *reinterpret_cast<const uint4*>(r + row * C + c0);
        const uint4 d4 = *reinterpret_cast<const uint4*>(dy + row * C + c0);
        const T* pa = reinterpret_cast<const T*>(&a);
        const T* pb = reinterpret_cast<const T*>(&b);
        const T* pd = reinterpret_cast<const T*>(&d4);
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const float t = to_f<T>(pa[e]) + (r ? to_f<T>(pb[e]) : 0.f);
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
        for (int e = 0; e < V; ++e) xh[pss][e] = dh[pss][e] = 0.f;
      }
    }
    s1 = wave_reduce_sum(s1) / C;
    s2 = wave_reduce_sum(s2) / C;
#pragma unroll
    for (int pss = 0; pss < NP; ++pss) {
      const int c0 = (pss * 64 + lane) * V;
      if (c0 < C) {
        uint4 o;
        T* po = reinterpret_cast<T*>(&o);
#pragma unroll
        for (int e = 0; e < V; ++e) po[e] = from_f<T>(rs * (dh[pss][e] - s1 - xh[pss][e] * s2));
        *reinterpret_cast<uint4*>(dxr + row * C + c0) = o;
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
  for (int c = threadIdx.x; c < C; c += 256) {
    atomicAdd(dgamma + c, sg[0][c] + sg[1][c] + sg[2][c] + sg[3][c]);
    atomicAdd(dbeta + c, sb[0][c] + sb[1][c] + sb[2][c] + sb[3][c]);
  }
}

// ---- cross entropy (sum) + gradient + top-k hits: one wave per row ----------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void ce_sum_kernel(const T* logits, const long* targets, T* dlogits, float* loss,
                                                     int* hits, long rows, int V, int topk, long ignore_index,
                                                     float dloss, float* row_loss, long ld) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* lr = logits + row * ld;
  const long tgt = targets[row];
  float mx = -INFINITY;
  for (int c = lane; c < V; c += 64) mx = fmaxf(mx, to_f<T>(lr[c]));
  mx = wave_reduce_max(mx);
  float se = 0.f;
  for (int c = lane; c < V; c += 64) se += expf(to_f<T>(lr[c]) - mx);
  se = wave_reduce_sum(se);
  const float lt = to_f<T>(lr[tgt]);
  const float lse = mx + logf(se);
  float gt = 0.f;
  for (int c = lane; c < V; c += 64) gt += (to_f<T>(lr[c]) > lt) ? 1.f : 0.f;
  gt = wave_reduce_sum(gt);
  if (dlogits) {
    T* dr = dlogits + row * ld;
    const float inv = dloss / se;
    for (int c = lane; c < V; c += 64) {
      float g = expf(to_f<T>(lr[c]) - mx) * inv;
      if (c == tgt) g -= dloss;
      dr[c] = from_f<T>(g);
    }
    for (long c = V + lane; c < ld; c += 64) dr[c] = from_f<T>(0.f);     // padding columns of a strided logits buffer
  }
  if (lane == 0) {
    if (loss) atomicAdd(loss, lse - lt);
    if (row_loss) row_loss[row] = lse - lt;
    if (hits && tgt != ignore_index) {
      atomicAdd(hits + 1, 1);
      if (gt < (float)topk) atomicAdd(hits, 1);
    }
  }
}

// The same arithmetic with the row held in registers: ONE pass over the logits (the kernel above reads a row four times with
// 2-byte loads and computes every exponential twice), rows handed out grid-stride so that a block ends with ONE atomic per
// counter instead of one per row (24576 same-address atomics per launch at the s1 shape: 328 us for 100 MB of traffic).
// NCH = values per lane: V <= 64 * NCH.
template <typename T, int NCH>
__global__ __launch_bounds__(256) void ce_sum_cached(const T* logits, const long* targets, T* dlogits, float* loss, int* hits,
                                                     long rows, int V, int topk, long ignore_index, float dloss,
                                                     float* row_loss, long ld) {
  __shared__ float s_loss[4];
  __shared__ int s_hit[4], s_cnt[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc_loss = 0.f;
  int acc_hit = 0, acc_cnt = 0;
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    const T* lr = logits + row * ld;
    const long tgt = targets[row];
    float v[NCH];
    float mx = -INFINITY, lt = -INFINITY;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = lane + 64 * j;
      v[j] = c < V ? to_f<T>(lr[c]) : -INFINITY;
      mx = fmaxf(mx, v[j]);
      if (c == tgt) lt = v[j];
    }
    mx = wave_reduce_max(mx);
    lt = wave_reduce_max(lt);                       // exactly one lane holds the target's logit
    float se = 0.f, gt = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      gt += v[j] > lt ? 1.f : 0.f;                  // columns >= V hold -inf: never counted
      v[j] = expf(v[j] - mx);                       // exp(-inf) = 0 for the columns >= V
      se += v[j];
    }
    se = wave_reduce_sum(se);
    gt = wave_reduce_sum(gt);
    const float lse = mx + logf(se);
    if (dlogits) {
      T* dr = dlogits + row * ld;
      const float inv = dloss / se;
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        const int c = lane + 64 * j;
        if (c < V) dr[c] = from_f<T>(v[j] * inv - (c == tgt ? dloss : 0.f));
      }
      for (long c = V + lane; c < ld; c += 64) dr[c] = from_f<T>(0.f);     // padding columns of a strided logits buffer
    }
    if (lane == 0) {
      acc_loss += lse - lt;
      if (row_loss) row_loss[row] = lse - lt;
      if (tgt != ignore_index) {
        ++acc_cnt;
        if (gt < (float)topk) ++acc_hit;
      }
    }
  }
  if (lane == 0) { s_loss[wave] = acc_loss; s_hit[wave] = acc_hit; s_cnt[wave] = acc_cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (loss) atomicAdd(loss, (s_loss[0] + s_loss[1]) + (s_loss[2] + s_loss[3]));
    if (hits) {
      const int c = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3], h = s_hit[0] + s_hit[1] + s_hit[2] + s_hit[3];
      if (c) atomicAdd(hits + 1, c);
      if (h) atomicAdd(hits, h);
    }
  }
}


// ---- ScaledAdam over a flat arena ----------------------------------------------------------------------------------
// chunk table: every block handles one chunk = a slice [begin, end) of ONE tensor
__global__ __launch_bounds__(256) void sa_stats_kernel(const float* p, const float* g, const evt_sa_chunk* chunks,
                                                       float* stats) {
  __shared__ float red[4];
  const evt_sa_chunk c = chunks[blockIdx.x];
  float pg = 0.f, pp = 0.f, gg = 0.f;
  for (long i = c.begin + threadIdx.x; i < c.end; i += 256) {
    const float a = p[i], b = g[i];
    pg += a * b; pp += a * a; gg += b * b;
  }
  pg = block_reduce_sum_256(pg, red);
  pp = block_reduce_sum_256(pp, red);
  gg = block_reduce_sum_256(gg, red);
  if (threadIdx.x == 0) {
    atomicAdd(stats + c.tensor * 3 + 0, pg);
    atomicAdd(stats + c.tensor * 3 + 1, pp);
    atomicAdd(stats + c.tensor * 3 + 2, gg);
  }
}

// per-tensor coefficient row: [0] scale_step*(1-beta1) (0 when no size update), [1] alpha = -lr*(1-beta1)*clamp(rms),
// [2] 1.0 for numel==1 tensors (plain-Adam branch), [3] unused
__global__ __launch_bounds__(256) void sa_apply_kernel(float* p, const float* g, float* delta, float* exp_avg_sq,
                                                       const evt_sa_chunk* chunks, const float* coef,
                                                       evt_scaled_adam_hp hp) {
  const evt_sa_chunk c = chunks[blockIdx.x];
  const float scale_term = coef[c.tensor * 4 + 0], alpha = coef[c.tensor * 4 + 1];
  const bool scalar = coef[c.tensor * 4 + 2] != 0.f;
  // The reference's gradient clipping reaches only the size-update statistics (scale_grads): _step_one_batch scales
  // a local copy (optim.py:462-464) while _step / _step_scalar re-read p.grad (:574, :609).  So no clip factor here.
  const float bc2 = 1.f - powf(hp.beta2, (float)(hp.step + 1));
  for (long i = c.begin + threadIdx.x; i < c.end; i += 256) {
    const float gr = g[i];
    float pv = p[i];
    float d = delta[i] * hp.beta1;
    float v = exp_avg_sq[i] * hp.beta2 + (1.f - hp.beta2) * gr * gr;
    exp_avg_sq[i] = v;
    if (!scalar) {
      d += pv * scale_term;
      const float vh = bc2 < 0.99f ? v / bc2 : v;
      d += gr / (sqrtf(vh) + hp.eps) * alpha;
      pv += d;
    } else {
      const float denom = sqrtf(v / bc2) + hp.eps;
      d += gr / denom * (-hp.lr * hp.scalar_lr_scale * (1.f - hp.beta1));
      pv = fminf(fmaxf(pv, -hp.scalar_max), hp.scalar_max);
      pv += d;
    }
    delta[i] = d;
    p[i] = pv;
  }
}

}  // namespace

extern "C" {

int evt_add_layernorm_fwd(int32_t dtype, const void* x, const void* r, const float* gamma, const float* beta, void* y,
                          float* mean, float* rstd, int64_t rows, int32_t C, float eps, void* stream) {
  if (!x || !gamma || !beta || !y || !mean || !rstd || rows <= 0 || C <= 0) return EVT_EINVAL;
  if (C > 1024 || C % 8) return EVT_ENOTSUP;
  hipStream_t st = (hipStream_t)stream;
  const int blocks = (int)((rows + 3) / 4);
#define LN_FWD(T, E) hipLaunchKernelGGL((add_ln_fwd<T, E>), dim3(blocks), dim3(256), 0, st, (const T*)x, (const T*)r, \
                                        gamma, beta, (T*)y, mean, rstd, (long)rows, C, eps)
  if (dtype == EVT_DT_HALF) { if (C <= 512) LN_FWD(h16_t, 1); else LN_FWD(h16_t, 2); }
  else if (dtype == EVT_DT_F32) { if (C <= 512) LN_FWD(float, 2); else LN_FWD(float, 4); }
  else return EVT_EINVAL;
#undef LN_FWD
  return evt_check_launch();
}

int evt_add_layernorm_bwd(int32_t dtype, const void* x, const void* r, const float* gamma, const void* dy,
                          const float* mean, const float* rstd, void* dxr, float* dgamma, float* dbeta, int64_t rows,
                          int32_t C, void* stream) {
  if (!x || !gamma || !dy || !mean || !rstd || !dxr || !dgamma || !dbeta || rows <= 0 || C <= 0) return EVT_EINVAL;
  if (C > 1024 || C % 8) return EVT_ENOTSUP;
  hipStream_t st = (hipStream_t)stream;
  long rpb = (rows + 1023) / 1024;
  if (rpb < 16) rpb = 16;
  const int blocks = (int)((rows + rpb - 1) / rpb);
#define LN_BWD(T, E) hipLaunchKernelGGL((add_ln_bwd<T, E>), dim3(blocks), dim3(256), 0, st, (const T*)x, (const T*)r, \
                                        gamma, (const T*)dy, mean, rstd, (T*)dxr, dgamma, dbeta, (long)rows, C, (int)rpb)
  if (dtype == EVT_DT_HALF) { if (C <= 512) LN_BWD(h16_t, 1); else LN_BWD(h16_t, 2); }
  else if (dtype == EVT_DT_F32) { if (C <= 512) LN_BWD(float, 2); else LN_BWD(float, 4); }
  else return EVT_EINVAL;
#undef LN_BWD
  return evt_check_launch();
}

static int launch_ce(int32_t dtype, const void* logits, const int64_t* targets, void* dlogits, float* loss,
                     float* row_loss, int32_t* hits, int64_t rows, int32_t V, int32_t topk, int64_t ignore_index,
                     float dloss, void* stream, int64_t ld = 0) {
  if (!logits || !targets || (!loss && !row_loss) || rows <= 0 || V <= 0) return EVT_EINVAL;
  if (ld == 0) ld = V;
  if (ld < V) return EVT_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (V <= 64 * 17) {                                // the s1 vocabulary (1025) and anything up to 1088 columns
    long nb = (rows + 3) / 4;
    if (nb > 2048) nb = 2048;
    if (dtype == EVT_DT_HALF)
      hipLaunchKernelGGL((ce_sum_cached<h16_t, 17>), dim3((int)nb), dim3(256), 0, st, (const h16_t*)logits,
                         (const long*)targets, (h16_t*)dlogits, loss, hits, (long)rows, V, topk, (long)ignore_index, dloss,
                         row_loss, (long)ld);
    else if (dtype == EVT_DT_F32)
      hipLaunchKernelGGL((ce_sum_cached<float, 17>), dim3((int)nb), dim3(256), 0, st, (const float*)logits,
                         (const long*)targets, (float*)dlogits, loss, hits, (long)rows, V, topk, (long)ignore_index, dloss,
                         row_loss, (long)ld);
    else return EVT_EINVAL;
    return evt_check_launch();
  }
  const int blocks = (int)((rows + 3) / 4);
  if (dtype == EVT_DT_HALF)
    hipLaunchKernelGGL(ce_sum_kernel<h16_t>, dim3(blocks), dim3(256), 0, st, (const h16_t*)logits,
                       (const long*)targets, (h16_t*)dlogits, loss, hits, (long)rows, V, topk, (long)ignore_index, dloss,
                       row_loss, (long)ld);
  else if (dtype == EVT_DT_F32)
    hipLaunchKernelGGL(ce_sum_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)logits, (const long*)targets,
                       (float*)dlogits, loss, hits, (long)rows, V, topk, (long)ignore_index, dloss, row_loss, (long)ld);
  else return EVT_EINVAL;
  return evt_check_launch();
}

int evt_ce_sum_fwd_bwd(int32_t dtype, const void* logits, const int64_t* targets, void* dlogits, float* loss,
                       int32_t* hits, int64_t rows, int32_t V, int32_t topk, int64_t ignore_index, float dloss,
                       void* stream) {
  if (!loss) return EVT_EINVAL;
  return launch_ce(dtype, logits, targets, dlogits, loss, nullptr, hits, rows, V, topk, ignore_index, dloss, stream);
}

int evt_ce_rows_fwd_bwd(int32_t dtype, const void* logits, const int64_t* targets, void* dlogits, float* row_loss,
                        int32_t* hits, int64_t rows, int32_t V, int32_t topk, int64_t ignore_index, void* stream) {
  if (!row_loss) return EVT_EINVAL;
  return launch_ce(dtype, logits, targets, dlogits, nullptr, row_loss, hits, rows, V, topk, ignore_index, 1.0f, stream);
}

int evt_ce_sum_fwd_bwd_ld(int32_t dtype, const void* logits, const int64_t* targets, void* dlogits, float* loss,
                          int32_t* hits, int64_t rows, int32_t V, int64_t ld, int32_t topk, int64_t ignore_index,
                          float dloss, void* stream) {
  if (!loss) return EVT_EINVAL;
  return launch_ce(dtype, logits, targets, dlogits, loss, nullptr, hits, rows, V, topk, ignore_index, dloss, stream, ld);
}

int evt_ce_rows_fwd_bwd_ld(int32_t dtype, const void* logits, const int64_t* targets, void* dlogits, float* row_loss,
                           int32_t* hits, int64_t rows, int32_t V, int64_t ld, int32_t topk, int64_t ignore_index,
                           void* stream) {
  if (!row_loss) return EVT_EINVAL;
  return launch_ce(dtype, logits, targets, dlogits, nullptr, row_loss, hits, rows, V, topk, ignore_index, 1.0f, stream, ld);
}

int evt_scaled_adam_stats(const float* param, const float* grad, const evt_sa_chunk* chunks, int32_t nchunks,
                          float* stats, void* stream) {
  if (!param || !grad || !chunks || nchunks <= 0 || !stats) return EVT_EINVAL;
  hipLaunchKernelGGL(sa_stats_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, param, grad, chunks, stats);
  return evt_check_launch();
}

int evt_scaled_adam_apply(float* param, const float* grad, float* delta, float* exp_avg_sq, const evt_sa_chunk* chunks,
                          int32_t nchunks, const float* coef, const evt_scaled_adam_hp* hp, void* stream) {
  if (!param || !grad || !delta || !exp_avg_sq || !chunks || nchunks <= 0 || !coef || !hp) return EVT_EINVAL;
  hipLaunchKernelGGL(sa_apply_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, param, grad, delta, exp_avg_sq,
                     chunks, coef, *hp);
  return evt_check_launch();
}

}  // extern "C"
