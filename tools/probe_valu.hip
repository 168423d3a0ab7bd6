// Hardware probe: issue rate of the VALU instructions the attention kernels are made of, on one SIMD with 1 / 2 / 4
// resident waves (gfx950).  Prints cycles per wave-instruction per SIMD.  Build: hipcc --offload-arch=gfx950 -O3
// tools/probe_valu.hip -o tools/probe_valu.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int OP>
__global__ void k(float* out, long long* cyc, int iters) {
  float a = threadIdx.x * 0.001f, b = 1.0001f, c = 0.5f, d = a + 1.f, e = a + 2.f, f = a + 3.f, g = a + 4.f, h = a + 5.f;
  unsigned u0 = threadIdx.x, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7;
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (OP == 0) { REP64(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %3, %3, %1, %2\n v_fma_f32 %4, %4, %1, %2\n v_fma_f32 %5, %5, %1, %2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));) }
    if (OP == 1) { REP64(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(a), "+v"(d), "+v"(e), "+v"(f));) }
    if (OP == 2) { REP64(asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %2, %2, %1\n v_lshrrev_b32 %3, 3, %3\n v_lshrrev_b32 %1, 1, %1" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
    if (OP == 3) { REP64(asm volatile("v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %2, %2, %1\n v_mul_u32_u24 %3, %3, %1\n v_mul_u32_u24 %0, %0, %2" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
    if (OP == 4) { REP64(asm volatile("v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_gt_u32 vcc, %1, %0\n v_cndmask_b32 %3, %3, %2, vcc" : "+v"(u0), "+v"(u1), "+v"(a), "+v"(d) :: "vcc");) }
    if (OP == 5) {
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 x = {a, d}, y = {e, f}, z = {g, h}, w = {b, c};
      REP64(asm volatile("v_pk_fma_f32 %0, %0, %3, %3\n v_pk_fma_f32 %1, %1, %3, %3\n v_pk_fma_f32 %2, %2, %3, %3\n v_pk_fma_f32 %0, %0, %3, %3" : "+v"(x), "+v"(y), "+v"(z), "+v"(w));)
      a = x[0] + y[0] + z[0]; d = x[1] + y[1] + z[1];
    }
    if (OP == 6) { REP64(asm volatile("v_max_f32 %0, %0, %1\n v_max_f32 %2, %2, %1\n v_add_f32 %3, %3, %1\n v_sub_f32 %4, %4, %1" : "+v"(a), "+v"(b), "+v"(d), "+v"(e), "+v"(f));) }
    if (OP == 7) { REP64(asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n v_cvt_pk_bf16_f32 %3, %1, %2\n v_cvt_pk_bf16_f32 %4, %1, %2\n v_cvt_pk_bf16_f32 %5, %1, %2" : "+v"(u0), "+v"(a), "+v"(d), "+v"(u1), "+v"(u2), "+v"(u3));) }
    if (OP == 8) { REP64(asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane32_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
  }
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h + u0 + u1 + u2 + u3;
  if ((threadIdx.x & 63) == 0) {   // span from the first wave's start to the LAST wave's end (the oldest wave has priority)
    atomicMin((unsigned long long*)&cyc[0], (unsigned long long)t0);
    atomicMax((unsigned long long*)&cyc[1], (unsigned long long)t1);
  }
}

template <int OP>
void run(const char* name, float* out, long long* cyc) {
  for (int waves : {1, 2, 4, 8}) {                      // waves per SIMD: block of waves*4 waves on one CU
    const int threads = waves * 4 * 64;
    if (threads > 1024) {                               // 8 waves/SIMD = two 1024-thread blocks on a CU is not forced; skip
      continue;
    }
    const int iters = 50;
    hipLaunchKernelGGL(k<OP>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long init[2] = {0x7fffffffffffffffLL, 0};
    hipMemcpy(cyc, init, sizeof(init), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k<OP>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long cc[2];
    hipMemcpy(cc, cyc, sizeof(cc), hipMemcpyDeviceToHost);
    const long long c = cc[1] - cc[0];
    const double n_inst = (double)iters * 64 * 4 * waves;   // wave-instructions issued on one SIMD
    printf("%-22s waves/SIMD %d: %.2f cycles per wave-instruction per SIMD (readcyclecounter ticks)\n", name, waves, (double)c / n_inst);
  }
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 1024 * 1024 * 4); hipMalloc(&cyc, 1024 * 8);
  // calibrate the counter: s_memtime ticks vs wall clock
  run<0>("v_fma_f32", out, cyc);
  run<5>("v_pk_fma_f32", out, cyc);
  run<6>("v_max/add/sub_f32", out, cyc);
  run<1>("v_exp_f32", out, cyc);
  run<2>("v_xor/v_lshrrev", out, cyc);
  run<3>("v_mul_u32_u24", out, cyc);
  run<4>("v_cmp+v_cndmask", out, cyc);
  run<7>("v_cvt_pk_bf16_f32", out, cyc);
  run<8>("v_permlane32/16_swap", out, cyc);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  long long init[2] = {0x7fffffffffffffffLL, 0};
  hipMemcpy(cyc, init, sizeof(init), hipMemcpyHostToDevice);
  hipEventRecord(e0); hipLaunchKernelGGL(k<0>, dim3(1), dim3(256), 0, 0, out, cyc, 2000); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); long long cc[2]; hipMemcpy(cc, cyc, 16, hipMemcpyDeviceToHost);
  printf("counter: %lld ticks in %.3f ms -> %.1f MHz\n", cc[1] - cc[0], ms, (cc[1] - cc[0]) / ms / 1e3);
  return 0;
}
