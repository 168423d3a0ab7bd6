// Micro-benchmark (round 6): how should 256 blocks of a split-K weight-gradient GEMM combine their 256 x 256 fp32 partial tiles?
//   mode 0: agent-scope atomicAdd into the shared tile (what wgrad_gemm does today)
//   mode 1: workgroup-scope (L2-local) atomicAdd into the slab of the XCD the block REALLY runs on (s_getreg XCC_ID)
//   mode 2: plain stores into a slab per block (deterministic reduce afterwards)
// build: hipcc --offload-arch=gfx950 -O3 tools/exp/probe_atomics.hip -o /tmp/probe_atomics ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int xcc_id() {
  int v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 15;
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, int ntile, float val, int* xcd_seen) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g = lane >> 4, wr = wave >> 2, wc = wave & 3;
  const int tile = blockIdx.x % ntile;
  const int x = xcc_id();
  if (tid == 0 && xcd_seen) xcd_seen[blockIdx.x] = x;
  float* base;
  if (MODE == 0) base = out + (size_t)tile * 65536;
  else if (MODE == 1) base = out + ((size_t)x * ntile + tile) * 65536;
  else base = out + (size_t)blockIdx.x * 65536;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* p = base + (wr * 128 + i * 16 + g * 4 + r) * 256 + wc * 64 + j * 16 + n;
        const float v = val + (float)(i + j + r) * 0.f;
        if (MODE == 0) atomicAdd(p, v);
        else if (MODE == 1) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else *p = v;
      }
}

int main() {
  const int nb = 256;
  float* out;
  int* seen;
  hipMalloc(&out, (size_t)nb * 65536 * 4);
  hipMalloc(&seen, nb * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int ntile : {4, 12, 16}) {
    for (int mode = 0; mode < 3; ++mode) {
      hipMemset(out, 0, (size_t)nb * 65536 * 4);
      auto launch = [&]() {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nb), dim3(512), 0, 0, out, ntile, 1.0f, seen);
        else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(nb), dim3(512), 0, 0, out, ntile, 1.0f, seen);
        else hipLaunchKernelGGL(k<2>, dim3(nb), dim3(512), 0, 0, out, ntile, 1.0f, seen);
      };
      launch();
      hipDeviceSynchronize();
      // correctness of the sums after ONE launch
      std::vector<float> h((size_t)(mode == 1 ? 8 * ntile : (mode == 0 ? ntile : nb)) * 65536);
      hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
      double bad = 0;
      if (mode != 2) {
        for (int t = 0; t < ntile; ++t) {
          const int want = nb / ntile + (t < nb % ntile ? 1 : 0);
          for (int e = 0; e < 65536; e += 257) {
            double s = 0;
            if (mode == 0) s = h[(size_t)t * 65536 + e];
            else for (int x = 0; x < 8; ++x) s += h[((size_t)x * ntile + t) * 65536 + e];
            if (s != want) bad += 1;
          }
        }
      }
      std::vector<int> hs(nb);
      hipMemcpy(hs.data(), seen, nb * 4, hipMemcpyDeviceToHost);
      int match = 0;
      for (int b = 0; b < nb; ++b) match += hs[b] == (b & 7);
      const int it = 20;
      hipEventRecord(e0);
      for (int i = 0; i < it; ++i) launch();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double us = ms * 1e3 / it;
      printf("ntile %2d mode %d: %7.1f us per launch, %6.2f TB/s of partial-tile bytes, wrong sums %g, blocks on xcd b%%8: %d/256\n",
             ntile, mode, us, (double)nb * 262144 / us / 1e6, bad, match);
    }
  }
  return 0;
}
