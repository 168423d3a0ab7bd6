// Input front-end of the s2 step, gfx950: the layout change from the reference's [B, C, T] tensors to channels-last
// rows, the frozen RVQ look-up, and the target-side mel projection.  None of it carries a gradient.
//
// Reference call sites (file:line under /root/reference):
//   spec -> enc_q.pre (1025 -> 192, 1x1)        src/easevoice/module/models.py:348-352  (the transposed, channel-padded copy
//                                                of the spectrogram is what lets the 1025-bin projection run on the
//                                                library's GEMM kernels: rows of 1025 values are not 16-byte aligned)
//   ssl_proj + quantizer (eval, n_q = 1)        src/easevoice/module/models.py:912-926, quantize.py:70-94,
//                                                core_vq.py:172-228 (EuclideanCodebook.quantize / dequantize)
//   spec_to_mel_torch                           src/easevoice/module/mel_processing.py:77-90, src/train/sovits.py:470-480
#include "evt_common.h"
#include "../../include/evt.h"

namespace {

// src fp32 [B][C][T] -> dst [B][T][Cp] (T = `dtype`), channels c >= C written as zeros.  32 x 32 tiles through LDS: loads
// run along t, stores along c.
template <typename T>
__global__ __launch_bounds__(256) void ncl_to_nlc_kernel(const float* src, T* dst, int C, int Tn, int Cp) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
  const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32, b = blockIdx.z;
  const float* s = src + (long)b * C * Tn;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, t = t0 + tx;
    tile[ty + 8 * i][tx] = (c < C && t < Tn) ? s[(long)c * Tn + t] : 0.f;
  }
  __syncthreads();
  T* d = dst + (long)b * Tn * Cp;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = t0 + ty + 8 * i, c = c0 + tx;
    if (t < Tn && c < Cp) d[(long)t * Cp + c] = from_f<T>(tile[tx][ty + 8 * i]);
  }
}

// ee[k] = |embed[k]|^2, one wave per code
__global__ __launch_bounds__(256) void rvq_norms_kernel(const float* embed, float* ee, int K, int D) {
  const int lane = threadIdx.x & 63, k = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= K) return;
  const float* e = embed + (long)k * D;
  float s = 0.f;
  for (int d = lane; d < D; d += 64) s += e[d] * e[d];
  s = wave_reduce_sum(s);
  if (lane == 0) ee[k] = s;
}

// one block per vector: dist[c] = -(|x|^2 - 2 x.e_c + |e_c|^2) (core_vq.py:172-180), first maximum wins; the code's vector
// is written `rep` times (the x2 nearest up-sampling of the 25 Hz codes, models.py:923-926)
__global__ __launch_bounds__(256) void rvq_select_kernel(const float* h, const float* dots, const float* embed, const float* ee,
                                                         int64_t* codes, float* q_out, int N, int D, int K, int rep) {
  __shared__ float red[4];
  __shared__ float bv[4];
  __shared__ int bi[4];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const float* x = h + (long)n * D;
  float xx = 0.f;
  for (int d = tid; d < D; d += 256) xx += x[d] * x[d];
  xx = block_reduce_sum_256(xx, red);
  float best = -INFINITY;
  int idx = 0x7fffffff;
  const float* dr = dots + (long)n * K;
  for (int c = tid; c < K; c += 256) {
    const float dist = -((xx - 2.f * dr[c]) + ee[c]);
    if (dist > best) { best = dist; idx = c; }       // c ascends per thread: the first maximum is kept
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(idx, o, 64);
    if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
  }
  if (lane == 0) { bv[w] = best; bi[w] = idx; }
  __syncthreads();
  best = bv[0]; idx = bi[0];
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (bv[i] > best || (bv[i] == best && bi[i] < idx)) { best = bv[i]; idx = bi[i]; }
  if (idx >= K) idx = 0;                               // all-NaN row: any valid code
  if (tid == 0) codes[n] = idx;
  const float* e = embed + (long)idx * D;
  for (int r = 0; r < rep; ++r) {
    float* o = q_out + ((long)n * rep + r) * D;
    for (int d = tid; d < D; d += 256) o[d] = e[d];
  }
}

// out[b][m][f] = log(max(sum_k basis[m][k] * spec[b][k][start_b + f], 1e-5)), f < nfr.  Block = 32 frames x 8 mels of one
// item (the training segment is 32 frames: B x 16 blocks fill the chip; one block per item took 340 us); the 8 basis rows
// sit in LDS, the 8 thread groups of a block split the bins (k = kp, kp + 8, ...) and read the spectrogram straight from
// global memory (32 consecutive frames per row: 128-byte runs), partial sums meet in LDS.
constexpr int SM_M = 8, SM_F = 32, SM_KP = 8;
__global__ __launch_bounds__(256) void spec_to_mel_kernel(const float* spec, const float* basis, const int64_t* starts,
                                                          float* out, int F, int Tn, int M, int nfr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* bs = reinterpret_cast<float*>(smem);             // [SM_M][F]
  float* part = bs + SM_M * F;                             // [SM_KP][SM_M][SM_F]
  const int tid = threadIdx.x, fx = tid & 31, kp = tid >> 5;
  const int b = blockIdx.y, f0 = blockIdx.x * SM_F, m0 = blockIdx.z * SM_M;
  const long st = starts ? (long)starts[b] : 0;
  for (int i = tid; i < SM_M * F; i += 256) {
    const int mm = i / F, kk = i - mm * F;
    bs[i] = m0 + mm < M ? basis[(long)(m0 + mm) * F + kk] : 0.f;
  }
  __syncthreads();
  const long t = st + f0 + fx;
  const bool ok = f0 + fx < nfr && t >= 0 && t < Tn;
  const float* s = spec + (long)b * F * Tn + (ok ? t : 0);
  float acc[SM_M];
#pragma unroll
  for (int i = 0; i < SM_M; ++i) acc[i] = 0.f;
  for (int k = kp; k < F; k += SM_KP) {
    const float v = ok ? s[(long)k * Tn] : 0.f;
#pragma unroll
    for (int i = 0; i < SM_M; ++i) acc[i] += bs[i * F + k] * v;
  }
#pragma unroll
  for (int i = 0; i < SM_M; ++i) part[(kp * SM_M + i) * SM_F + fx] = acc[i];
  __syncthreads();
  {
    const int mm = tid >> 5;                               // 8 mels x 32 frames = 256 outputs, one per thread
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < SM_KP; ++q) v += part[(q * SM_M + mm) * SM_F + fx];
    if (f0 + fx < nfr && m0 + mm < M) out[((long)b * M + m0 + mm) * nfr + f0 + fx] = logf(fmaxf(v, 1e-5f));
  }
}

}  // namespace

extern "C" {

int evt_ncl_to_nlc(int32_t dtype, const float* src, void* dst, int32_t B, int32_t C, int32_t T, int32_t Cp, void* stream) {
  if (!src || !dst || B <= 0 || C <= 0 || T <= 0 || Cp < C) return EVT_EINVAL;
  if (dtype != EVT_DT_F32 && dtype != EVT_DT_HALF) return EVT_EINVAL;
  const dim3 grid((T + 31) / 32, (Cp + 31) / 32, B);
  evt_set_last_tag("ncl_to_nlc");
  if (dtype == EVT_DT_HALF)
    hipLaunchKernelGGL(ncl_to_nlc_kernel<h16_t>, grid, dim3(256), 0, (hipStream_t)stream, src, (h16_t*)dst, C, T, Cp);
  else
    hipLaunchKernelGGL(ncl_to_nlc_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, src, (float*)dst, C, T, Cp);
  return evt_check_launch();
}

int evt_rvq_norms(const float* embed, float* ee, int32_t K, int32_t D, void* stream) {
  if (!embed || !ee || K <= 0 || D <= 0) return EVT_EINVAL;
  evt_set_last_tag("rvq_norms");
  hipLaunchKernelGGL(rvq_norms_kernel, dim3((K + 3) / 4), dim3(256), 0, (hipStream_t)stream, embed, ee, K, D);
  return evt_check_launch();
}

int evt_rvq_select(const float* h, const float* dots, const float* embed, const float* ee, int64_t* codes, float* q_out,
                   int32_t N, int32_t D, int32_t K, int32_t rep, void* stream) {
  if (!h || !dots || !embed || !ee || !codes || !q_out || N <= 0 || D <= 0 || K <= 0 || rep <= 0) return EVT_EINVAL;
  evt_set_last_tag("rvq_select");
  hipLaunchKernelGGL(rvq_select_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, h, dots, embed, ee, codes, q_out, N, D,
                     K, rep);
  return evt_check_launch();
}

int evt_spec_to_mel(const float* spec, const float* basis, const int64_t* starts, float* out, int32_t B, int32_t F,
                    int32_t T, int32_t M, int32_t nfr, void* stream) {
  if (!spec || !basis || !out || B <= 0 || F <= 0 || T <= 0 || M <= 0 || nfr <= 0) return EVT_EINVAL;
  evt_set_last_tag("spec_to_mel");
  const dim3 grid((nfr + SM_F - 1) / SM_F, B, (M + SM_M - 1) / SM_M);
  const size_t lds = ((size_t)SM_M * F + SM_KP * SM_M * SM_F) * sizeof(float);
  if (lds > 64 * 1024) return EVT_ENOTSUP;                 // 2048-point STFT: 41 KB
  hipLaunchKernelGGL(spec_to_mel_kernel, grid, dim3(256), lds, (hipStream_t)stream, spec, basis, starts, out, F, T, M, nfr);
  return evt_check_launch();
}

}  // extern "C"
