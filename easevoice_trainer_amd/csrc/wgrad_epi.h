// Epilogue of the MFMA weight-gradient kernels (conv_deep.hip: wgrad_deep / wgrad_ring, wgrad_halo.hip): where a block's
// partial tile goes.
//
// Positions (the GEMM's K) are split over blockIdx.y.  Two ways to combine the partial tiles of the splits:
//   * classic  (WgP::parts == 0): fp32 atomics into the one gradient image.  Order of the additions = order in which the
//     blocks finish, so two runs differ in the last bits, and every partial tile crosses the fabric as 64-byte atomic
//     packets (PMC, round 2: 45.6 MB of atomic writes per wgrad_deep launch against 27.5 MB of operands).
//   * slabs    (WgP::parts  > 0): split s owns slab s of the image -- slab 0 is the classic image, slabs 1.. live in an
//     extra buffer -- and writes it with plain 16-byte stores: `+=` into slabs that already hold sums of this step
//     (WgP::dirty0 for slab 0, WgP::prev_used for the others), `=` into the rest: the usual case, no read at all.  Every (tile, slab) has exactly one owner block and launches are stream-ordered, so the bits do not depend
//     on timing; evt_wn_grad_multi adds the slabs in index order while it reads the image anyway.  The tile leaves through LDS as whole runs (one dy-channel x
//     all taps x 32 x-channels = NT * 128 contiguous bytes of the image).
#pragma once
#include "conv_p.h"

namespace evt_conv {

__device__ __forceinline__ float* wg_slab(const WgP& p, int s) {
  return s == 0 ? p.dw : p.dw_extra + (long)(s - 1) * p.part_stride;
}

// Block tile = (2 wave rows x MI x 16) dy-channels x (NT taps x 32 x-channels); wave (wr, wc) holds, per (i, t), the
// MFMA tile of rows wr*16*MI + i*16 + g8*4 + r and columns wc*16 + j16.  `scratch` = at least 32 * (NT*32 + 4) floats of
// LDS that no wave still reads (callers barrier before).  One pass per row tile i: 32 image rows at a time.
template <int MI, int NT>
__device__ __forceinline__ void wg_store_slab(const WgP& p, float* slab, bool add, float* scratch,
                                              const f32x4 (&acc)[MI][NT], int ntap, int a0, int ch, int t0, int wr, int wc,
                                              int g8, int j16) {
  constexpr int PITCH = NT * 32 + 4;                       // floats; rows stay 16-byte aligned, g8 groups hit other banks
  constexpr int CPR = NT * 8;                              // 16-byte chunks per row
  const int tid = threadIdx.x;
  if constexpr (NT == 1) {
    // one tap: a row of the tile is 128 bytes and the 16 lanes of an MFMA column group already cover 64 contiguous bytes
    // of it -- two barriers per pass would cost more than they save on these 8-10 us launches
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int a = a0 + wr * 16 * MI + i * 16 + g8 * 4 + r;
        float* dst = slab + (((long)a * p.nchunk + ch) * p.KHp + t0) * 32 + wc * 16 + j16;
        *dst = add ? *dst + acc[i][0][r] : acc[i][0][r];
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) scratch[(wr * 16 + g8 * 4 + r) * PITCH + t * 32 + wc * 16 + j16] = acc[i][t][r];
    __syncthreads();
    for (int c = tid; c < 32 * CPR; c += 256) {
      const int lr = c / CPR, cc = c - lr * CPR;
      if (cc >= ntap * 8) continue;
      const int a = a0 + (lr >> 4) * 16 * MI + i * 16 + (lr & 15);
      float* dst = slab + (((long)a * p.nchunk + ch) * p.KHp + t0) * 32 + cc * 4;
      f32x4 v = *reinterpret_cast<const f32x4*>(scratch + lr * PITCH + cc * 4);
      if (add) {
        const f32x4 o = *reinterpret_cast<const f32x4*>(dst);
        v += o;
      }
      *reinterpret_cast<f32x4*>(dst) = v;
    }
  }
}

// the common tail of the kernels: classic atomics or the slab store; `smem` is the block's whole dynamic LDS
template <int MI, int NT>
__device__ __forceinline__ void wg_finish(const WgP& p, unsigned char* smem, const f32x4 (&acc)[MI][NT], int ntap, int a0,
                                          int ch, int t0, int wr, int wc, int g8, int j16, int split) {
  if (p.parts > 0) {
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) p.used[0] = p.now_used;
    wg_store_slab<MI, NT>(p, wg_slab(p, split), split == 0 ? p.dirty0 != 0 : split < p.prev_used, reinterpret_cast<float*>(smem), acc, ntap,
                          a0, ch, t0, wr, wc, g8, j16);
    return;
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t >= ntap) continue;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int a = a0 + wr * 16 * MI + i * 16 + g8 * 4 + r;
        const long off = (((long)a * p.nchunk + ch) * p.KHp + t0 + t) * 32 + wc * 16 + j16;
        atomicAdd(p.dw + off, acc[i][t][r]);
      }
  }
}

// The fused bias gradient = column sums of the A operand (dy) over the block's positions, ON THE MATRIX PIPE: one extra
// MFMA per A fragment against an all-ones B fragment; every column of the 16 x 16 result then holds the fragment's 16
// channel sums (the j16 == 0 lanes keep them).  Reading the staged tile back instead -- 64 two-byte LDS loads per thread
// and stage -- made the blocks that own a bias the slowest of the grid: the s1 dense-layer gradient ran at 605 TFLOP/s
// with and 795 without it (round 4).  The two waves holding the same A fragments (wc = 0 / 1) take the even / odd ones.
template <int MI>
__device__ __forceinline__ void wg_bias_mma(f32x4 (&bacc)[MI], const h16x8 (&a)[MI], int wc) {
  h16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (evt_hn)1.0f;
#pragma unroll
  for (int i = 0; i < MI; ++i)
    if ((i & 1) == wc) bacc[i] = EVT_MFMA_16x16x32(a[i], ones, bacc[i], 0, 0, 0);
}

__device__ __forceinline__ void wg_finish_bias(const WgP& p, int channel, float bsum, int split);

template <int MI>
__device__ __forceinline__ void wg_finish_bias_mma(const WgP& p, const f32x4 (&bacc)[MI], int a0, int wr, int wc, int g8,
                                                   int j16, int split) {
  if (j16 != 0) return;
#pragma unroll
  for (int i = 0; i < MI; ++i)
    if ((i & 1) == wc)
#pragma unroll
      for (int r = 0; r < 4; ++r) wg_finish_bias(p, a0 + wr * 16 * MI + i * 16 + g8 * 4 + r, bacc[i][r], split);
}

// fused bias gradient (column sums of dy over this block's positions): atomics into the parameter's gradient, or --
// slab mode -- plain stores into db_part[split][channel], summed by evt_wn_grad_multi
__device__ __forceinline__ void wg_finish_bias(const WgP& p, int channel, float bsum, int split) {
  if (p.parts > 0 && p.db_part) {
    float* d = p.db_part + (long)split * p.CA + channel;
    *d = split < p.prev_used ? *d + bsum : bsum;
    if (channel == 0 && split == 0) p.used[1] = p.now_used;
  } else {
    atomicAdd(p.dbias + channel, bsum);
  }
}

}  // namespace evt_conv
