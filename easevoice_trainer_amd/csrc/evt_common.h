// Shared device helpers for the evt HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bfloat16 bits; all bf16 tensors cross the C ABI as uint16 storage

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define EVT_OK 0
#define EVT_EINVAL 22
#define EVT_ENOTSUP 95
#define EVT_ELAUNCH 5

#define EVT_DT_F32 0
#define EVT_DT_BF16 1

#define EVT_ACT_NONE 0
#define EVT_ACT_LRELU 1
#define EVT_ACT_TANH 2

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even, NaN preserved (same rounding torch uses for float -> bfloat16).  The native __bf16 cast lets
// the compiler emit gfx950's v_cvt_pk_bf16_f32 (two conversions per instruction) instead of six integer ops per value.
__device__ __forceinline__ bf16_t f2bf(float f) {
  const __bf16 h = (__bf16)f;
  return __builtin_bit_cast(bf16_t, h);
}

template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<bf16_t>(bf16_t v) { return bf2f(v); }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float v) { return f2bf(v); }

__device__ __forceinline__ float lrelu_f(float v, float slope) { return v > 0.f ? v : v * slope; }

// derivative factor of an activation, expressed through the activation's OUTPUT value `ya`
// (lrelu with slope>0 keeps the sign, tanh' = 1 - tanh^2)
__device__ __forceinline__ float dact_from_out(int kind, float ya, float slope) {
  if (kind == EVT_ACT_LRELU) return ya > 0.f ? 1.f : slope;
  if (kind == EVT_ACT_TANH) return 1.f - ya * ya;
  return 1.f;
}

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum for blockDim.x == 256 (4 waves); `red` is a 4-float LDS scratch
__device__ __forceinline__ float block_reduce_sum_256(float v, float* red) {
  v = wave_reduce_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// name of the kernel instantiation the last entry point launched on this host thread (bench.py's roofline leg reads
// it right after a call to attribute HIP-event timings to the same names rocprofv3 reports)
extern "C" void evt_set_last_tag(const char* fmt, ...);

static inline int evt_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? EVT_OK : EVT_ELAUNCH;
}
