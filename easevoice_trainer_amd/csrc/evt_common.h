// Shared device helpers for the evt HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The library is built twice from these sources (easevoice_trainer_amd/build.py): libevt_hip.so with bfloat16 as its 16-bit
// floating type and libevt_hip_f16.so (-DEVT_HALF_F16) with IEEE half -- the reference's `fp16_run` autocast type
// (src/train/sovits.py:459-525).  A build serves EVT_DT_F32 and ITS half code (EVT_DT_HALF); the other half code is
// EVT_ENOTSUP.  Everything below the C ABI is written against `h16_t` (raw 16-bit storage), `evt_hn` (the native
// arithmetic type of that storage) and the conversions h2f / f2h; the 16-bit LDS transposes, LDS-DMA and fragment
// layouts are the same for both types, the MFMA opcode differs (EVT_MFMA_16x16x32).
typedef uint16_t h16_t;  // raw bits of the build's 16-bit float; all 16-bit tensors cross the C ABI as uint16 storage

typedef __attribute__((ext_vector_type(4))) float f32x4;

#define EVT_OK 0
#define EVT_EINVAL 22
#define EVT_ENOTSUP 95
#define EVT_ELAUNCH 5

#define EVT_DT_F32 0
#define EVT_DT_BF16 1
#define EVT_DT_F16 2

#ifdef EVT_HALF_F16
typedef _Float16 evt_hn;
#define EVT_DT_HALF EVT_DT_F16
#define EVT_HALF_NAME "f16"
#define EVT_MFMA_16x16x32(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z)
#else
typedef __bf16 evt_hn;
#define EVT_DT_HALF EVT_DT_BF16
#define EVT_HALF_NAME "bf16"
#define EVT_MFMA_16x16x32(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z)
#endif
typedef evt_hn h16x8 __attribute__((ext_vector_type(8)));

#define EVT_ACT_NONE 0
#define EVT_ACT_LRELU 1
#define EVT_ACT_TANH 2

// 16-bit -> fp32: exact for both types.  bfloat16 is the upper half of the fp32 pattern (one shift); half goes through
// the native conversion (v_cvt_f32_f16).
__device__ __forceinline__ float h2f(h16_t v) {
#ifdef EVT_HALF_F16
  return (float)__builtin_bit_cast(_Float16, v);
#else
  return __uint_as_float(((uint32_t)v) << 16);
#endif
}

// fp32 -> 16-bit, round-to-nearest-even, NaN preserved (the rounding torch uses for float -> bfloat16 / float16; a half
// result beyond 65504 is +-inf, which is what the GradScaler of the fp16 mode looks for).  The native casts let the
// compiler emit gfx950's v_cvt_pk_bf16_f32 / v_cvt_pkrtz-free v_cvt_f16_f32 instead of integer sequences.
__device__ __forceinline__ h16_t f2h(float f) {
  const evt_hn h = (evt_hn)f;
  return __builtin_bit_cast(h16_t, h);
}

// the two values of a packed pair (one dword of a 16-bit tensor: element 2i in the low half, 2i + 1 in the high half)
__device__ __forceinline__ float h2f_lo(uint32_t d) {
#ifdef EVT_HALF_F16
  return h2f((h16_t)(d & 0xFFFFu));
#else
  return __uint_as_float(d << 16);
#endif
}
__device__ __forceinline__ float h2f_hi(uint32_t d) {
#ifdef EVT_HALF_F16
  return h2f((h16_t)(d >> 16));
#else
  return __uint_as_float(d & 0xFFFF0000u);
#endif
}
__device__ __forceinline__ uint32_t f2h_pack(float lo, float hi) { return (uint32_t)f2h(lo) | ((uint32_t)f2h(hi) << 16); }

template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<h16_t>(h16_t v) { return h2f(v); }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ h16_t from_f<h16_t>(float v) { return f2h(v); }

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
