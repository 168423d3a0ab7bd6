// fold_partials: second stage of the reductions whose output is tiny and whose input is long -- the weight gradients of
// the Cout = 1 / Cin = 1 layers and of the 16- / 32-channel vocoder stages (src/easevoice/module/models.py:419-446,
// 490-497, 536, 566-574 through torch.autograd).  Hundreds of blocks each hold a partial copy of a 0.2 - 12 KB result.
// Added with fp32 atomics, every address takes one atomic per block, and same-address atomics from different XCDs
// serialise at the memory side (~50 ns each: 256 blocks = 13 us per launch whatever the kernel did before, and an order
// of additions that changes from run to run).  Instead: every block stores its partial in a scratch row, and this kernel
// adds the rows in a fixed order:  out[e] += sum_b part[b * stride + e].
//   block = 16 consecutive outputs x 16 row groups; a thread adds rows g, g + 16, g + 32, ... (8 loads in flight), the 16
//   groups meet in LDS and are added in index order.  256 rows of 3072 floats: 192 blocks, two round trips per thread.
#include "conv_p.h"

namespace evt_conv {
namespace {

__global__ __launch_bounds__(256) void fold_partials(const float* part, long stride, int nb, float* out, long n) {
  __shared__ float red[16][17];
  const int el = threadIdx.x & 15, g = threadIdx.x >> 4;
  const long e = (long)blockIdx.x * 16 + el;
  float s = 0.f;
  if (e < n) {
    int b = g;
    for (; b + 7 * 16 < nb; b += 8 * 16) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = part[(long)(b + u * 16) * stride + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += t[u];
    }
    for (; b < nb; b += 16) s += part[(long)b * stride + e];
  }
  red[g][el] = s;
  __syncthreads();
  if (g == 0 && e < n) {
    float v = red[0][el];
#pragma unroll
    for (int k = 1; k < 16; ++k) v += red[k][el];
    out[e] += v;
  }
}

}  // namespace

int launch_fold_partials(const float* part, long stride, int nb, float* out, long n, hipStream_t st) {
  if (!part || !out || nb <= 0 || n <= 0) return EVT_EINVAL;
  hipLaunchKernelGGL(fold_partials, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, st, part, stride, nb, out, n);
  return evt_check_launch();
}

}  // namespace evt_conv
