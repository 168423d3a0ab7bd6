// Shared device helpers of the fused HiFi-GAN ResBlock kernels (resunit.hip: forward, resunit_bwd.hip: backward) for the
// narrow vocoder stages, gfx950 bf16.
//
// LDS rows.  A wave keeps the rows of its 64-position unit in a private LDS area, 2C bytes per row, no padding:
//   C = 16: 32-byte rows.  C = 32: 64-byte rows, the 16-byte slot s of row r stored at slot s ^ (2 * ((r >> 2) & 1)).
// Both are conflict-free for the two access patterns of the kernels, at any row shift (taps are row shifts):
//   * MFMA B operand of a convolution: lane (n, g) reads 16 bytes of row n + shift (ds_read_b128);
//   * transposing reads of the weight gradients: a 16-lane group reads 4 rows x 32 bytes (ds_read_b64_tr_b16); a lane
//     group g takes rows {4g..4g+3} and {16+4g..16+4g+3} of a 32-position block, so that the 8 rows of one read of a
//     half-wave are 8 consecutive rows = one 256-byte bank window.  The permutation of K this implies is the same for
//     both MFMA operands, so the sum over positions is unchanged.
//
// Scheduling.  hipcc leaves the fully unrolled loops of these kernels in source order -- fragment read, s_waitcnt,
// MFMA, next read -- and with one or two waves per SIMD nothing hides the LDS round trip (measured: 144 waits per unit,
// 47-57 us for 42 MB).  conv_stage / the weight-gradient loop are therefore software-pipelined by hand: the
// fragments of step s+1 are requested before the MFMAs of step s, and an empty `asm volatile` with a memory clobber
// ("tie") pins the request where it was written; the waits the compiler inserts in front of a tie are counted
// (lgkmcnt(n) with the younger requests still in flight).
#pragma once
#include "evt_common.h"

namespace evt_ru {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

__device__ __forceinline__ void tie(u32x4& v) { asm volatile("" : "+v"(v)::"memory"); }
__device__ __forceinline__ void tie(u32x2& v) { asm volatile("" : "+v"(v)::"memory"); }
__device__ __forceinline__ h16x8 as_h8(const u32x4& v) { return __builtin_bit_cast(h16x8, v); }

// byte offset of the 16-byte piece `pc` of row `row` inside a region
template <int CI> __device__ __forceinline__ int piece_off(int row, int pc) {
  if constexpr (CI == 16) return row * 32 + pc * 16;
  else return row * 64 + ((pc ^ (((row >> 2) & 1) << 1)) << 4);
}
// byte offset of channel c (multiple of 4) of row `row`
template <int CI> __device__ __forceinline__ int chan_off(int row, int c) {
  return piece_off<CI>(row, c >> 3) + (c & 7) * 2;
}
// byte offset a transposing read of rows row.. takes for the 16-channel tile `tile`, lane column bytes colb (0, 8, 16, 24)
template <int CI> __device__ __forceinline__ int tr_off(int row, int tile, int colb) {
  if constexpr (CI == 16) return row * 32 + colb;
  else return (row * 64 + tile * 32 + colb) ^ (((row >> 2) & 1) << 5);
}

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_p;

// the two transposing reads of one MFMA operand fragment: rows r..r+3 and r+16..r+19 of 16 channels -> lo/hi halves
template <int PITCH>
__device__ __forceinline__ u32x4 tr_frag(const unsigned char* p) {
  union { struct { s16x4 lo, hi; } h; u32x4 v; } r;
  r.h.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(p));
  r.h.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(p + 16 * PITCH));
  return r.v;
}

__device__ __forceinline__ float sum8(const u32x4& v) {          // sum of the eight bf16 of a fragment
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += h2f_lo(v[i]) + h2f_hi(v[i]);
  return s;
}

__device__ __forceinline__ uint32_t lrelu2(uint32_t d, float slope) {      // two packed bf16
  float a = h2f_lo(d), b = h2f_hi(d);
  a = a > 0.f ? a : a * slope;
  b = b > 0.f ? b : b * slope;
  return f2h_pack(a, b);
}
__device__ __forceinline__ uint4 lrelu8(uint4 v, float slope) {
  return make_uint4(lrelu2(v.x, slope), lrelu2(v.y, slope), lrelu2(v.z, slope), lrelu2(v.w, slope));
}
__device__ __forceinline__ uint32_t scale2(uint32_t d, float s) {      // two packed bf16 times s
  const float a = h2f_lo(d) * s, b = h2f_hi(d) * s;
  return f2h_pack(a, b);
}
__device__ __forceinline__ uint4 scale8(uint4 v, float s) {
  return make_uint4(scale2(v.x, s), scale2(v.y, s), scale2(v.z, s), scale2(v.w, s));
}

// One convolution of a unit as MFMAs: acc[i][j] += W tile i x rows tile j over NK K-steps of 32 (= padded taps x CI / 32).
//   wl:   the weight image in LDS, rows = output channels, pitch WPITCH bytes, K index contiguous;
//   rows: the LDS region of the input rows; the row of (position tile j, lane position n, tap) is n + 16 j + tap * dil.
// Lane (n = lane & 15, g = lane >> 4) holds K elements ks * 32 + g * 8 ..+7 of both operands.
template <int CI, int NK, int MT, int NT>
__device__ __forceinline__ void conv_stage(f32x4 (&acc)[MT][NT], const unsigned char* wl, const unsigned char* rows,
                                           const int dil, const int n, const int g) {
  constexpr int PITCH = CI * 2;
  constexpr int WPITCH = NK * 64 + 16;
  u32x4 fa[2][MT], fb[2][NT];
  auto issue = [&](const int ks, const int s) {
    const int kk = ks * 32 + g * 8;
    const int tap = kk / CI, ci = kk - tap * CI;
#pragma unroll
    for (int i = 0; i < MT; ++i) fa[s][i] = *reinterpret_cast<const u32x4*>(wl + (i * 16 + n) * WPITCH + kk * 2);
    const unsigned char* base = rows + piece_off<CI>(n + tap * dil, ci >> 3);
#pragma unroll
    for (int j = 0; j < NT; ++j) fb[s][j] = *reinterpret_cast<const u32x4*>(base + j * 16 * PITCH);
  };
  issue(0, 0);
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) {
    const int s = ks & 1;
    if (ks + 1 < NK) issue(ks + 1, s ^ 1);
#pragma unroll
    for (int i = 0; i < MT; ++i) tie(fa[s][i]);
#pragma unroll
    for (int j = 0; j < NT; ++j) tie(fb[s][j]);
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int i = 0; i < MT; ++i)
        acc[i][j] = EVT_MFMA_16x16x32(as_h8(fa[s][i]), as_h8(fb[s][j]), acc[i][j], 0, 0, 0);
  }
}

// both weight images of a unit (prepared [rows = CI][KTOT] bf16) -> LDS, pitch KTOT * 2 + 16 bytes; whole block
template <int CI, int KTOT>
__device__ __forceinline__ void load_weights(unsigned char* wl1, unsigned char* wl2, const h16_t* w1, const h16_t* w2) {
  constexpr int WPITCH = KTOT * 2 + 16;
  for (int idx = threadIdx.x; idx < CI * (KTOT / 8); idx += 256) {
    const int co = idx / (KTOT / 8), part = idx - co * (KTOT / 8);
    *reinterpret_cast<uint4*>(wl1 + co * WPITCH + part * 16) = *reinterpret_cast<const uint4*>(w1 + (long)co * KTOT + part * 8);
    *reinterpret_cast<uint4*>(wl2 + co * WPITCH + part * 16) = *reinterpret_cast<const uint4*>(w2 + (long)co * KTOT + part * 8);
  }
}

}  // namespace evt_ru
