// rows16_gemm: y[M][N] = x[M][K] . W[N][K]^T (+ bias) for M <= 16 rows, bf16, gfx950 -- the 1x1 "convolutions" applied to
// one vector per batch item: the weight-normed conditioning layers of the WN stacks (ge [B, 512] -> [B, 2*192*layers],
// src/easevoice/module/modules.py:168-176 of the reference) and the vocoder's cond layer (models.py:446,455).
// The implicit-GEMM conv kernels tile POSITIONS; with 16 positions they ran 37-74 us per launch.  Here the weight rows are
// the MFMA's M side: a workgroup owns 16 output columns, its four waves split K (every fourth K step of 32) and combine
// through LDS; each lane streams its own 16-byte pieces of the weight row straight from global memory.
// Forward takes the REG image (for k = 1: W row-major [N][K]), backward-data the ALT image (W^T row-major [K][N]).
#include "conv_p.h"
#include <cstdlib>

namespace {

__global__ __launch_bounds__(256) void rows16_gemm(const h16_t* __restrict__ x, const h16_t* __restrict__ w,
                                                   const float* __restrict__ bias, h16_t* __restrict__ y, int M, int N,
                                                   int K) {
  __shared__ f32x4 red[3][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int col0 = blockIdx.x * 16;
  const h16_t* wr = w + (long)(col0 + n) * K + g * 8;            // A: weight row col0 + n
  const h16_t* xr = x + (long)n * K + g * 8;                      // B: input row n (zero beyond M)
  const bool live = n < M;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int nks = K / 32;
  for (int ks = wave; ks < nks; ks += 4) {
    const h16x8 a = *reinterpret_cast<const h16x8*>(wr + ks * 32);
    h16x8 b;
    if (live) b = *reinterpret_cast<const h16x8*>(xr + ks * 32);
    else { union { uint4 u; h16x8 v; } z; z.u = make_uint4(0, 0, 0, 0); b = z.v; }
    acc = EVT_MFMA_16x16x32(a, b, acc, 0, 0, 0);
  }
  if (wave > 0) red[wave - 1][lane] = acc;
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const f32x4 o = red[i][lane];
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] += o[r];
    }
    // lane: output columns col0 + g*4 .. +3 of input row n
    if (live) {
      h16_t o4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) o4[r] = f2h(acc[r] + (bias ? bias[col0 + g * 4 + r] : 0.f));
      *reinterpret_cast<uint2*>(y + (long)n * N + col0 + g * 4) = *reinterpret_cast<uint2*>(o4);
    }
  }
}

}  // namespace

namespace evt_conv {

// k = 1, dense, un-fused, at most 16 rows in total: true when rows16 takes the launch
bool rows16_eligible(const evt_conv1d_params* c, int rows, int n_out, int k_red, bool fused) {
  static const bool off = getenv("EVT_NO_ROWS16") != nullptr;
  return !off && !fused && c->dtype == EVT_DT_HALF && c->impl == EVT_IMPL_AUTO && c->k == 1 && c->stride == 1 &&
         c->groups == 1 && !c->transposed && c->pad == 0 && rows <= 16 && n_out % 16 == 0 && k_red % 32 == 0;
}

int launch_rows16(const void* x, const void* w, const float* bias, void* y, int M, int N, int K, hipStream_t st) {
  evt_set_last_tag("rows16_gemm<bf16>");
  hipLaunchKernelGGL(rows16_gemm, dim3(N / 16), dim3(256), 0, st, (const h16_t*)x, (const h16_t*)w, bias, (h16_t*)y, M, N, K);
  return evt_check_launch();
}

}  // namespace evt_conv
