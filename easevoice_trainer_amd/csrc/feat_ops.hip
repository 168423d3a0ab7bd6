// Element-wise / normalisation kernels of the feature extractors that produce the training set's 3-bert/*.pt and
// 4-cnhubert/*.pt (src/normalization/normalize.py:88-106,132-180 of the reference, which runs
// transformers' BertForMaskedLM and HubertModel on the CPU).  The transformer layers of both models run on the library's
// convolution / attention / LayerNorm kernels; what those models need beyond them is here:
//   * GELU (erf form: transformers' "gelu" activation of both models), with an optional bias row added first and an
//     optional time-axis trim -- the tail of HuBERT's positional convolution (HubertPositionalConvEmbedding: conv with
//     padding k/2, drop the last frame of an even kernel, GELU) is ONE launch;
//   * per-channel normalisation over time + GELU: HuBERT's first feature-extractor layer (GroupNorm with one channel per
//     group, "feat_extract_norm": "group", then GELU) -- statistics per (item, channel) over all frames, two passes for
//     the variance (a 10 s clip has 32 000 frames: E[x^2] - E[x]^2 in fp32 would lose the variance of a near-constant
//     channel).
// Inference only: forward launches, no saved statistics.  HBM-bound streams: 16-byte accesses along the channel axis.
#include "evt_common.h"
#include "../../include/evt.h"

namespace {

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }

// out[b][t][c] = gelu(x[b][t][c] + bias[c]) for t < t_out <= t_in (rows t >= t_out of the input are dropped)
template <typename T>
__global__ void gelu_rows_kernel(const T* x, const float* bias, T* out, long nseq, int t_in, int t_out, int C) {
  const long n = nseq * (long)t_out * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long row = i / C;
    const int c = (int)(i - row * C);
    const long b = row / t_out;
    const long t = row - b * t_out;
    const float v = to_f<T>(x[(b * t_in + t) * C + c]) + (bias ? bias[c] : 0.f);
    out[i] = from_f<T>(gelu_f(v));
  }
}

// one block per (item, 64 channels): lane = channel, the four waves stride over the frames
template <typename T>
__global__ __launch_bounds__(256) void channel_norm_gelu_kernel(const T* x, const float* gamma, const float* beta, float eps,
                                                                T* out, int Tn, int C, int gelu) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + lane;
  const bool live = c < C;
  const T* xb = x + (long)blockIdx.x * Tn * C;
  T* ob = out + (long)blockIdx.x * Tn * C;
  float s = 0.f;
  if (live)
    for (int t = wave; t < Tn; t += 4) s += to_f<T>(xb[(long)t * C + c]);
  red[wave][lane] = s;
  __syncthreads();
  const float mean = (red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]) / (float)Tn;
  __syncthreads();
  float q = 0.f;
  if (live)
    for (int t = wave; t < Tn; t += 4) {
      const float d = to_f<T>(xb[(long)t * C + c]) - mean;
      q += d * d;
    }
  red[wave][lane] = q;
  __syncthreads();
  const float var = (red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]) / (float)Tn;     // biased, as GroupNorm
  const float rstd = rsqrtf(var + eps);
  if (!live) return;
  const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
  for (int t = wave; t < Tn; t += 4) {
    const float v = (to_f<T>(xb[(long)t * C + c]) - mean) * rstd * g + bt;
    ob[(long)t * C + c] = from_f<T>(gelu ? gelu_f(v) : v);
  }
}

inline int ew_blocks(long n) {
  long b = (n + 255) / 256;
  return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" {

int evt_gelu_rows_fwd(int32_t dtype, const void* x, const float* bias, void* out, int64_t nseq, int32_t t_in, int32_t t_out,
                      int32_t C, void* stream) {
  if (!x || !out || nseq <= 0 || t_out <= 0 || t_out > t_in || C <= 0) return EVT_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const long n = nseq * (long)t_out * C;
  if (dtype == EVT_DT_HALF)
    hipLaunchKernelGGL(gelu_rows_kernel<h16_t>, dim3(ew_blocks(n)), dim3(256), 0, st, (const h16_t*)x, bias, (h16_t*)out,
                       (long)nseq, t_in, t_out, C);
  else if (dtype == EVT_DT_F32)
    hipLaunchKernelGGL(gelu_rows_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, st, (const float*)x, bias, (float*)out,
                       (long)nseq, t_in, t_out, C);
  else return EVT_ENOTSUP;
  return evt_check_launch();
}

int evt_channel_norm_gelu_fwd(int32_t dtype, const void* x, const float* gamma, const float* beta, float eps, void* out,
                              int32_t nseq, int32_t T, int32_t C, int32_t apply_gelu, void* stream) {
  if (!x || !out || nseq <= 0 || T <= 0 || C <= 0 || !(eps >= 0.f)) return EVT_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(nseq, (C + 63) / 64);
  if (dtype == EVT_DT_HALF)
    hipLaunchKernelGGL(channel_norm_gelu_kernel<h16_t>, grid, dim3(256), 0, st, (const h16_t*)x, gamma, beta, eps, (h16_t*)out,
                       T, C, apply_gelu);
  else if (dtype == EVT_DT_F32)
    hipLaunchKernelGGL(channel_norm_gelu_kernel<float>, grid, dim3(256), 0, st, (const float*)x, gamma, beta, eps, (float*)out,
                       T, C, apply_gelu);
  else return EVT_ENOTSUP;
  return evt_check_launch();
}

}  // extern "C"
