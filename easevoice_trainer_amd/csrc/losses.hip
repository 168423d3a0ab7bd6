// Masked KL term of the s2 generator loss and the workspace-size query of the C ABI (gfx950 only).
//
// kl_loss, src/easevoice/module/losses.py:46-61:
//   kl = logs_p - logs_q - 0.5 + 0.5 * (z_p - m_p)^2 * exp(-2 logs_p);   loss = sum(kl * z_mask) / sum(z_mask)
// z_mask is the sequence mask [B, 1, T] (1 for t < len[b]), so sum(z_mask) = sum_b len[b] (frames, not frames x channels).
// The reference runs this as ~12 element-wise launches and two full reductions over [B, 192, T]; here it is one
// streaming pass each way.  HBM-bound: 4 tensors read once forward (and once more + 4 written backward).
#include "evt_common.h"
#include "../../include/evt.h"

namespace {

__device__ __forceinline__ float ld_any(const void* p, int dt, long i) {
  return dt == EVT_DT_HALF ? h2f(reinterpret_cast<const h16_t*>(p)[i]) : reinterpret_cast<const float*>(p)[i];
}
__device__ __forceinline__ void st_any(void* p, int dt, long i, float v) {
  if (dt == EVT_DT_HALF) reinterpret_cast<h16_t*>(p)[i] = f2h(v); else reinterpret_cast<float*>(p)[i] = v;
}

struct KlArgs {
  const void* zp; const void* lq; const void* mp; const void* lp;
  void* dzp; void* dlq; void* dmp; void* dlp;
  int dt_zp, dt_lq, dt_mp, dt_lp;
  const int* lens;
  int B, T, C, time_inner;   // time_inner: element (b, c, t) at (b*C + c)*T + t; else (b, t, c) at (b*T + t)*C + c
};

__device__ __forceinline__ bool kl_live(const KlArgs& a, long i) {
  if (!a.lens) return true;
  long b, t;
  if (a.time_inner) { t = i % a.T; b = i / ((long)a.T * a.C); }
  else { const long row = i / a.C; b = row / a.T; t = row - b * a.T; }
  return t < a.lens[b];
}

__global__ __launch_bounds__(256) void masked_kl_fwd_kernel(KlArgs a, long n, float* out) {
  __shared__ float red[4];
  float acc = 0.f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    if (!kl_live(a, i)) continue;
    const float zp = ld_any(a.zp, a.dt_zp, i), lq = ld_any(a.lq, a.dt_lq, i), mp = ld_any(a.mp, a.dt_mp, i),
                lp = ld_any(a.lp, a.dt_lp, i);
    const float d = zp - mp;
    acc += lp - lq - 0.5f + 0.5f * d * d * __expf(-2.f * lp);
  }
  acc = block_reduce_sum_256(acc, red);
  if (threadIdx.x == 0) atomicAdd(out, acc);
  if (blockIdx.x == 0 && threadIdx.x == 0) {        // out[1] = sum(z_mask): live frames
    long cnt = 0;
    for (int b = 0; b < a.B; ++b) cnt += a.lens ? min(max(a.lens[b], 0), a.T) : a.T;
    out[1] = (float)cnt;
  }
}

__global__ __launch_bounds__(256) void masked_kl_bwd_kernel(KlArgs a, long n, const float* dloss, const float* count) {
  const float g = dloss[0] / count[0];
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float gz = 0.f, glq = 0.f, glp = 0.f;
    if (kl_live(a, i)) {
      const float zp = ld_any(a.zp, a.dt_zp, i), mp = ld_any(a.mp, a.dt_mp, i), lp = ld_any(a.lp, a.dt_lp, i);
      const float d = zp - mp, e = __expf(-2.f * lp);
      gz = g * d * e;
      glq = -g;
      glp = g * (1.f - d * d * e);
    }
    if (a.dzp) st_any(a.dzp, a.dt_zp, i, gz);
    if (a.dlq) st_any(a.dlq, a.dt_lq, i, glq);
    if (a.dmp) st_any(a.dmp, a.dt_mp, i, -gz);
    if (a.dlp) st_any(a.dlp, a.dt_lp, i, glp);
  }
}

int kl_args(KlArgs& a, const evt_kl_params* p, const void* zp, const void* lq, const void* mp, const void* lp,
            const int32_t* lens) {
  if (!p || !zp || !lq || !mp || !lp || p->B <= 0 || p->T <= 0 || p->C <= 0) return EVT_EINVAL;
  const int dts[4] = {p->dt_z_p, p->dt_logs_q, p->dt_m_p, p->dt_logs_p};
  for (int d : dts) if (d != EVT_DT_F32 && d != EVT_DT_HALF) return EVT_EINVAL;
  a.zp = zp; a.lq = lq; a.mp = mp; a.lp = lp;
  a.dzp = a.dlq = a.dmp = a.dlp = nullptr;
  a.dt_zp = dts[0]; a.dt_lq = dts[1]; a.dt_mp = dts[2]; a.dt_lp = dts[3];
  a.lens = lens; a.B = p->B; a.T = p->T; a.C = p->C; a.time_inner = p->time_inner ? 1 : 0;
  return EVT_OK;
}

}  // namespace

extern "C" {

int evt_masked_kl_fwd(const evt_kl_params* p, const void* z_p, const void* logs_q, const void* m_p, const void* logs_p,
                      const int32_t* lens, float* out2, void* stream) {
  KlArgs a;
  if (int rc = kl_args(a, p, z_p, logs_q, m_p, logs_p, lens)) return rc;
  if (!out2) return EVT_EINVAL;
  const long n = (long)p->B * p->T * p->C;
  long blocks = (n + 256L * 8 - 1) / (256L * 8);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(masked_kl_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, n, out2);
  return evt_check_launch();
}

int evt_masked_kl_bwd(const evt_kl_params* p, const void* z_p, const void* logs_q, const void* m_p, const void* logs_p,
                      const int32_t* lens, const float* dloss, const float* count, void* dz_p, void* dlogs_q, void* dm_p,
                      void* dlogs_p, void* stream) {
  KlArgs a;
  if (int rc = kl_args(a, p, z_p, logs_q, m_p, logs_p, lens)) return rc;
  if (!dloss || !count) return EVT_EINVAL;
  a.dzp = dz_p; a.dlq = dlogs_q; a.dmp = dm_p; a.dlp = dlogs_p;
  const long n = (long)p->B * p->T * p->C;
  long blocks = (n + 256L * 8 - 1) / (256L * 8);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(masked_kl_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, n, dloss, count);
  return evt_check_launch();
}

int64_t evt_workspace_bytes(int32_t op, const int64_t* dims, int32_t ndims) {
  if (!dims && ndims > 0) return -1;
  switch (op) {
    case EVT_WS_MEL:           // dims: nseq, wav_len, n_fft, hop, n_mels
      if (ndims != 5) return -1;
      return 4 * evt_mel_workspace_floats((int32_t)dims[0], (int32_t)dims[1], (int32_t)dims[2], (int32_t)dims[3],
                                          (int32_t)dims[4]);
    case EVT_WS_ATTN_BWD:      // dims: B, H, L -> delta_ws fp32 [B][H][L]
    case EVT_WS_MHA_BWD:   // dims: B, H, Tq -> delta_ws fp32 [B*H][Tq]
      if (ndims != 3) return -1;
      return 4 * dims[0] * dims[1] * dims[2];
    case EVT_WS_MASKED_KL:     // out2: (sum, live frames)
      return 8;
    case EVT_WS_NONE:
      return 0;
    default:
      return -1;
  }
}

}  // extern "C"
