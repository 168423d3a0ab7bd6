// Flash attention for the s1 text->semantic GPT with the ANALYTIC prefix-LM + key-padding mask (gfx950).
//
// Replaces F.scaled_dot_product_attention(q, k, v, attn_mask) at
// src/easevoice/soundstorm/auto_reg/modules/patched_mha_with_cache.py:452-454 and the materialised float mask
// [B*16, L, L] built at src/easevoice/soundstorm/auto_reg/models/t2s_model.py:456-479 (2 GiB at B=32, L=1024):
//   key j is visible from query i  <=>  j is not a padded key  AND  ( j < x_len  if i < x_len  else  j <= i )
//   padded keys: text columns j >= x_lens[b], audio columns j - x_len >= y_lens[b]; padded QUERY rows still attend.
//
// head_dim D = 32 means one 16x16x32 bf16 MFMA per 16x16 score tile.  All three bf16 kernels keep the softmax
// statistics of a query in the lane that owns the query's MFMA column:
//   forward / dQ: S^T = K Q^T (rows = keys, cols = queries); a lane holds 8 keys of ONE query, row max/sum need two
//     cross-lane steps (xor 16, 32); P^T stays in registers and is directly the B operand of O^T += V^T P^T
//     (dQ^T += K^T dS^T), whose A operand comes from the row-major LDS tile through ds_read_b64_tr_b16;
//   dK/dV: S = Q K^T (rows = queries, cols = keys); P / dS are directly the B operands of dV^T += dO^T P and
//     dK^T += Q^T dS.  No score tile ever goes through LDS or HBM, nothing is accumulated with atomics.
// The fp32 kernels (one thread per row) are the parity path: exact-order fp32, no MFMA.
#include "evt_common.h"
#include "../../include/evt.h"

namespace {

struct AP {
  const void* q; const void* k; const void* v; const void* o; const void* d_o;
  void* out; void* dq; void* dk; void* dv;
  float* lse; const float* delta;
  const int* x_lens; const int* y_lens;
  int B, L, H, x_len;
  long sb, sl, sh;     // q/k/v/dq/dk/dv element strides
  long ob, ol, oh;     // o / d_o element strides
  float scale;
  unsigned drop_thr;   // keep iff hash >= thr ; 0 = no dropout
  unsigned seed;
  float keep_scale;    // 1 / (1 - p)
};

// Counter-based dropout mask (attention dropout of SDPA, patched_mha_with_cache.py:452-454): a pure function of
// (seed, b*H+h, query, key) so forward and both backward kernels regenerate the same mask with no storage.
// Only 24-bit multiplies (full rate on CDNA; 32-bit integer multiplies are quarter rate) and xor-shifts.
__device__ __forceinline__ unsigned drop_row(const AP& p, unsigned bh, int qi) {
  const unsigned a = (bh << 11) ^ (unsigned)qi;                    // < 2^24 for B*H <= 8192, L <= 2048
  return __umul24(a & 0xFFFFFFu, 0x9E3779u) ^ p.seed ^ (a >> 7);
}
__device__ __forceinline__ float drop_mult2(const AP& p, unsigned row, int kj) {
  unsigned x = row + __umul24((unsigned)kj, 0x85EBCBu);
  x ^= x >> 15;
  x = __umul24(x & 0xFFFFFFu, 0x2C1B3Du) ^ (x >> 9);
  x ^= x << 13;
  x ^= x >> 17;
  return x >= p.drop_thr ? p.keep_scale : 0.f;
}
__device__ __forceinline__ float drop_mult(const AP& p, unsigned bh, int qi, int kj) {
  if (p.drop_thr == 0u) return 1.f;
  return drop_mult2(p, drop_row(p, bh, qi), kj);
}

__device__ __forceinline__ bool visible(int qi, int kj, int x_len, int xl, int yl) {
  const bool key_ok = kj < x_len ? (kj < xl) : (kj - x_len < yl);
  if (!key_ok) return false;
  return qi < x_len ? (kj < x_len) : (kj <= qi);
}

__device__ __forceinline__ bf16x8 tr2(const bf16_t* p0, const bf16_t* p1) {
  const unsigned a0 = (unsigned)(uintptr_t)p0, a1 = (unsigned)(uintptr_t)p1;
  uint2 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(lo), "=&v"(hi) : "v"(a0), "v"(a1) : "memory");
  union { uint4 u; bf16x8 v; } r;
  r.u = make_uint4(lo.x, lo.y, hi.x, hi.y);
  return r.v;
}

__device__ __forceinline__ bf16x8 pack8(const float* p) {
  union { bf16x8 v; bf16_t e[8]; } r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.e[i] = f2bf(p[i]);
  return r.v;
}

__device__ __forceinline__ bf16x8 ld8(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }

constexpr int D = 32;
constexpr int PITCH = D + 8;   // LDS row pitch in elements (80 B: 16-byte aligned, not a power of two)

// k-slot e of lane group g inside a 32-wide block: e<4 -> g*4+e ; e>=4 -> 16 + g*4 + (e-4)
__device__ __forceinline__ int slot32(int g, int e) { return e < 4 ? g * 4 + e : 16 + g * 4 + (e - 4); }

// ---------------------------------------------------------------------------------------------------------
// tile classification (wave-uniform): a 16-query x 32-key (or 32-query x 16-key) tile is
//   FULL  : every (q, k) pair visible -> no per-element mask arithmetic
//   EMPTY : no pair visible          -> the tile is skipped, MFMAs included
//   MIXED : evaluate visible() per element
// ---------------------------------------------------------------------------------------------------------
enum { TILE_EMPTY = 0, TILE_FULL = 1, TILE_MIXED = 2 };

__device__ __forceinline__ int classify(int q0, int q1, int k0, int k1, int L, int x_len, int xl, int yl) {
  // q in [q0, q1], k in [k0, k1] (inclusive)
  if (k0 >= L || q0 >= L) return TILE_EMPTY;
  // keys: padding
  int key_state;  // 0 none visible, 1 all unpadded, 2 mixed
  if (k1 < x_len) key_state = k1 < xl ? 1 : (k0 >= xl ? 0 : 2);
  else if (k0 >= x_len) key_state = (k1 - x_len < yl) ? 1 : ((k0 - x_len >= yl) ? 0 : 2);
  else key_state = 2;
  if (key_state == 0) return TILE_EMPTY;
  // causal / prefix structure
  int c_state;
  if (q1 < x_len) c_state = k1 < x_len ? 1 : (k0 >= x_len ? 0 : 2);          // text rows see text keys only
  else if (q0 >= x_len) c_state = k1 <= q0 ? 1 : (k0 > q1 ? 0 : 2);          // audio rows see keys <= row
  else c_state = 2;
  if (c_state == 0) return TILE_EMPTY;
  if (k1 >= L || q1 >= L) return TILE_MIXED;
  return (key_state == 1 && c_state == 1) ? TILE_FULL : TILE_MIXED;
}

__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }

// all-reduce over the four lanes {n, n+16, n+32, n+48} that hold one query's scores, with the gfx950 VALU lane-swap
// instructions instead of LDS-pipe shuffles: v_permlane32_swap exchanges the upper half of one operand with the lower
// half of the other, v_permlane16_swap exchanges odd 16-lane rows with even rows.
__device__ __forceinline__ float quad_max(float v) {
  unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  u = __float_as_uint(v);
  auto r2 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r2[0]), __uint_as_float(r2[1]));
}
__device__ __forceinline__ float quad_sum(float v) {
  unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  u = __float_as_uint(v);
  auto r2 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
}
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

// ---------------------------------------------------------------------------------------------------------
// forward (bf16): block = 4 waves x 2 query tiles (128 queries) of one (b, h); key blocks of 64 through LDS
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_fwd_bf16(AP p) {
  __shared__ __attribute__((aligned(16))) bf16_t Ks[64 * PITCH];
  __shared__ __attribute__((aligned(16))) bf16_t Vs[64 * PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int b = blockIdx.y / p.H, h = blockIdx.y % p.H;
  const int qblk = blockIdx.x * 128;
  const int xl = p.x_lens[b], yl = p.y_lens[b];
  const bf16_t* Q = reinterpret_cast<const bf16_t*>(p.q) + b * p.sb + h * p.sh;
  const bf16_t* K = reinterpret_cast<const bf16_t*>(p.k) + b * p.sb + h * p.sh;
  const bf16_t* V = reinterpret_cast<const bf16_t*>(p.v) + b * p.sb + h * p.sh;
  const float sc2 = p.scale * LOG2E;

  bf16x8 qf[2];
  f32x4 ot[2][2];
  float m[2], l[2];
  int qi[2], qt0[2];
  unsigned drow[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    qt0[t] = qblk + wave * 32 + t * 16;
    qi[t] = qt0[t] + n;
    const int qc = min(qi[t], p.L - 1);
    qf[t] = ld8(Q + qc * p.sl + g * 8);
    ot[t][0] = ot[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    m[t] = -INFINITY;
    l[t] = 0.f;
    drow[t] = drop_row(p, blockIdx.y, qi[t]);
  }
  const int qlast = min(qblk + 127, p.L - 1);
  const int kmax = (qlast < p.x_len) ? p.x_len : qlast + 1;
  for (int kb = 0; kb < kmax; kb += 64) {
    __syncthreads();
    {
      const int r = tid >> 2, c8 = tid & 3;
      const int kj = min(kb + r, p.L - 1);
      *reinterpret_cast<uint4*>(Ks + r * PITCH + c8 * 8) = *reinterpret_cast<const uint4*>(K + kj * p.sl + c8 * 8);
      *reinterpret_cast<uint4*>(Vs + r * PITCH + c8 * 8) = *reinterpret_cast<const uint4*>(V + kj * p.sl + c8 * 8);
    }
    __syncthreads();
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
      const int k0 = kb + sb * 32;
      int cls[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) cls[t] = classify(qt0[t], qt0[t] + 15, k0, k0 + 31, p.L, p.x_len, xl, yl);
      if (cls[0] == TILE_EMPTY && cls[1] == TILE_EMPTY) continue;
      const bf16x8 ka0 = ld8(Ks + (sb * 32 + n) * PITCH + g * 8);
      const bf16x8 ka1 = ld8(Ks + (sb * 32 + 16 + n) * PITCH + g * 8);
      const bf16_t* vrow = Vs + (sb * 32 + g * 4 + (n >> 2)) * PITCH + 4 * (n & 3);
      const bf16x8 va0 = tr2(vrow, vrow + 16 * PITCH);
      const bf16x8 va1 = tr2(vrow + 16, vrow + 16 * PITCH + 16);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (cls[t] == TILE_EMPTY) continue;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const f32x4 s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka0, qf[t], z, 0, 0, 0);
        const f32x4 s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka1, qf[t], z, 0, 0, 0);
        float s[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = (e < 4 ? s0[e] : s1[e - 4]) * sc2;
        if (cls[t] == TILE_MIXED) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int kj = k0 + slot32(g, e);
            if (!(kj < p.L && visible(qi[t], kj, p.x_len, xl, yl))) s[e] = -INFINITY;
          }
        }
        float mx = fmaxf(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])), fmaxf(fmaxf(s[4], s[5]), fmaxf(s[6], s[7])));
        mx = quad_max(mx);
        const float mn = fmaxf(m[t], mx);
        const bool dead = mn == -INFINITY;                 // nothing visible yet for this query
        const float alpha = dead ? 1.f : fexp2(m[t] - mn);
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] = dead ? 0.f : fexp2(s[e] - mn); sum += s[e]; }
        sum = quad_sum(sum);
        l[t] = l[t] * alpha + sum;
        m[t] = mn;
        if (p.drop_thr) {
#pragma unroll
          for (int e = 0; e < 8; ++e) s[e] *= drop_mult2(p, drow[t], k0 + slot32(g, e));
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { ot[t][0][r] *= alpha; ot[t][1][r] *= alpha; }
        const bf16x8 pf = pack8(s);
        ot[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va0, pf, ot[t][0], 0, 0, 0);
        ot[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va1, pf, ot[t][1], 0, 0, 0);
      }
    }
  }
  bf16_t* O = reinterpret_cast<bf16_t*>(p.out) + b * p.ob + h * p.oh;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (qi[t] >= p.L) continue;
    const float inv = l[t] > 0.f ? 1.f / l[t] : 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      bf16_t o4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) o4[r] = f2bf(ot[t][mt][r] * inv);
      *reinterpret_cast<uint2*>(O + qi[t] * p.ol + mt * 16 + g * 4) = *reinterpret_cast<uint2*>(o4);
    }
    if (g == 0) p.lse[((long)b * p.H + h) * p.L + qi[t]] = (m[t] + __log2f(l[t])) * LN2;
  }
}

// ---------------------------------------------------------------------------------------------------------
// dQ (bf16): same ownership as forward; dQ^T += K^T dS^T
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_dq_bf16(AP p) {
  __shared__ __attribute__((aligned(16))) bf16_t Ks[64 * PITCH];
  __shared__ __attribute__((aligned(16))) bf16_t Vs[64 * PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int b = blockIdx.y / p.H, h = blockIdx.y % p.H;
  const int qblk = blockIdx.x * 128;
  const int xl = p.x_lens[b], yl = p.y_lens[b];
  const bf16_t* Q = reinterpret_cast<const bf16_t*>(p.q) + b * p.sb + h * p.sh;
  const bf16_t* K = reinterpret_cast<const bf16_t*>(p.k) + b * p.sb + h * p.sh;
  const bf16_t* V = reinterpret_cast<const bf16_t*>(p.v) + b * p.sb + h * p.sh;
  const bf16_t* dO = reinterpret_cast<const bf16_t*>(p.d_o) + b * p.ob + h * p.oh;
  const float sc2 = p.scale * LOG2E;
  bf16x8 qf[2], dof[2];
  f32x4 dqt[2][2];
  float lse2[2], dl[2];
  int qi[2], qt0[2];
  unsigned drow[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    qt0[t] = qblk + wave * 32 + t * 16;
    qi[t] = qt0[t] + n;
    const int qc = min(qi[t], p.L - 1);
    qf[t] = ld8(Q + qc * p.sl + g * 8);
    dof[t] = ld8(dO + qc * p.ol + g * 8);
    lse2[t] = p.lse[((long)b * p.H + h) * p.L + qc] * LOG2E;
    dl[t] = p.delta[((long)b * p.H + h) * p.L + qc];
    dqt[t][0] = dqt[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    drow[t] = drop_row(p, blockIdx.y, qi[t]);
  }
  const int qlast = min(qblk + 127, p.L - 1);
  const int kmax = (qlast < p.x_len) ? p.x_len : qlast + 1;
  for (int kb = 0; kb < kmax; kb += 64) {
    __syncthreads();
    {
      const int r = tid >> 2, c8 = tid & 3;
      const int kj = min(kb + r, p.L - 1);
      *reinterpret_cast<uint4*>(Ks + r * PITCH + c8 * 8) = *reinterpret_cast<const uint4*>(K + kj * p.sl + c8 * 8);
      *reinterpret_cast<uint4*>(Vs + r * PITCH + c8 * 8) = *reinterpret_cast<const uint4*>(V + kj * p.sl + c8 * 8);
    }
    __syncthreads();
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
      const int k0 = kb + sb * 32;
      int cls[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) cls[t] = classify(qt0[t], qt0[t] + 15, k0, k0 + 31, p.L, p.x_len, xl, yl);
      if (cls[0] == TILE_EMPTY && cls[1] == TILE_EMPTY) continue;
      const bf16x8 ka0 = ld8(Ks + (sb * 32 + n) * PITCH + g * 8);
      const bf16x8 ka1 = ld8(Ks + (sb * 32 + 16 + n) * PITCH + g * 8);
      const bf16x8 va0 = ld8(Vs + (sb * 32 + n) * PITCH + g * 8);
      const bf16x8 va1 = ld8(Vs + (sb * 32 + 16 + n) * PITCH + g * 8);
      const bf16_t* krow = Ks + (sb * 32 + g * 4 + (n >> 2)) * PITCH + 4 * (n & 3);
      const bf16x8 kt0 = tr2(krow, krow + 16 * PITCH);
      const bf16x8 kt1 = tr2(krow + 16, krow + 16 * PITCH + 16);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (cls[t] == TILE_EMPTY) continue;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const f32x4 s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka0, qf[t], z, 0, 0, 0);
        const f32x4 s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka1, qf[t], z, 0, 0, 0);
        const f32x4 d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va0, dof[t], z, 0, 0, 0);
        const f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va1, dof[t], z, 0, 0, 0);
        float ds[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int kj = k0 + slot32(g, e);
          float pr = fexp2((e < 4 ? s0[e] : s1[e - 4]) * sc2 - lse2[t]);
          if (cls[t] == TILE_MIXED && !(kj < p.L && visible(qi[t], kj, p.x_len, xl, yl))) pr = 0.f;
          float dp = (e < 4 ? d0[e] : d1[e - 4]);
          if (p.drop_thr) dp *= drop_mult2(p, drow[t], kj);
          ds[e] = pr * (dp - dl[t]) * p.scale;
        }
        const bf16x8 dsf = pack8(ds);
        dqt[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt0, dsf, dqt[t][0], 0, 0, 0);
        dqt[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt1, dsf, dqt[t][1], 0, 0, 0);
      }
    }
  }
  bf16_t* dQ = reinterpret_cast<bf16_t*>(p.dq) + b * p.sb + h * p.sh;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (qi[t] >= p.L) continue;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      bf16_t o4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) o4[r] = f2bf(dqt[t][mt][r]);
      *reinterpret_cast<uint2*>(dQ + qi[t] * p.sl + mt * 16 + g * 4) = *reinterpret_cast<uint2*>(o4);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// dK/dV (bf16): block = 4 waves x 2 key tiles (128 keys); query blocks of 64 through LDS
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_dkv_bf16(AP p) {
  __shared__ __attribute__((aligned(16))) bf16_t Qs[64 * PITCH];
  __shared__ __attribute__((aligned(16))) bf16_t Os[64 * PITCH];
  __shared__ float lse_s[64], dl_s[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int b = blockIdx.y / p.H, h = blockIdx.y % p.H;
  const int kblk = blockIdx.x * 128;
  const int xl = p.x_lens[b], yl = p.y_lens[b];
  const bf16_t* Q = reinterpret_cast<const bf16_t*>(p.q) + b * p.sb + h * p.sh;
  const bf16_t* K = reinterpret_cast<const bf16_t*>(p.k) + b * p.sb + h * p.sh;
  const bf16_t* V = reinterpret_cast<const bf16_t*>(p.v) + b * p.sb + h * p.sh;
  const bf16_t* dO = reinterpret_cast<const bf16_t*>(p.d_o) + b * p.ob + h * p.oh;
  const float sc2 = p.scale * LOG2E;
  bf16x8 kf[2], vf[2];
  f32x4 dkt[2][2], dvt[2][2];
  int kj[2], kt0[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    kt0[t] = kblk + wave * 32 + t * 16;
    kj[t] = kt0[t] + n;
    const int kc = min(kj[t], p.L - 1);
    kf[t] = ld8(K + kc * p.sl + g * 8);
    vf[t] = ld8(V + kc * p.sl + g * 8);
    dkt[t][0] = dkt[t][1] = dvt[t][0] = dvt[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int q_begin = (kblk >= p.x_len) ? (kblk / 64) * 64 : 0;
  for (int qb = q_begin; qb < p.L; qb += 64) {
    __syncthreads();
    {
      const int r = tid >> 2, c8 = tid & 3;
      const int qq = min(qb + r, p.L - 1);
      *reinterpret_cast<uint4*>(Qs + r * PITCH + c8 * 8) = *reinterpret_cast<const uint4*>(Q + qq * p.sl + c8 * 8);
      *reinterpret_cast<uint4*>(Os + r * PITCH + c8 * 8) = *reinterpret_cast<const uint4*>(dO + qq * p.ol + c8 * 8);
      if (tid < 64) {
        const int q2 = min(qb + tid, p.L - 1);
        lse_s[tid] = p.lse[((long)b * p.H + h) * p.L + q2] * LOG2E;
        dl_s[tid] = p.delta[((long)b * p.H + h) * p.L + q2];
      }
    }
    __syncthreads();
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
      const int q0 = qb + sb * 32;
      int cls[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) cls[t] = classify(q0, q0 + 31, kt0[t], kt0[t] + 15, p.L, p.x_len, xl, yl);
      if (cls[0] == TILE_EMPTY && cls[1] == TILE_EMPTY) continue;
      const bf16x8 qa0 = ld8(Qs + (sb * 32 + n) * PITCH + g * 8);
      const bf16x8 qa1 = ld8(Qs + (sb * 32 + 16 + n) * PITCH + g * 8);
      const bf16x8 oa0 = ld8(Os + (sb * 32 + n) * PITCH + g * 8);
      const bf16x8 oa1 = ld8(Os + (sb * 32 + 16 + n) * PITCH + g * 8);
      const bf16_t* qrow = Qs + (sb * 32 + g * 4 + (n >> 2)) * PITCH + 4 * (n & 3);
      const bf16_t* orow = Os + (sb * 32 + g * 4 + (n >> 2)) * PITCH + 4 * (n & 3);
      const bf16x8 qt0 = tr2(qrow, qrow + 16 * PITCH), qt1 = tr2(qrow + 16, qrow + 16 * PITCH + 16);
      const bf16x8 dt0 = tr2(orow, orow + 16 * PITCH), dt1 = tr2(orow + 16, orow + 16 * PITCH + 16);
      float lq[8], dq8[8];
      unsigned drw[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        lq[e] = lse_s[sb * 32 + slot32(g, e)];
        dq8[e] = dl_s[sb * 32 + slot32(g, e)];
        drw[e] = p.drop_thr ? drop_row(p, blockIdx.y, q0 + slot32(g, e)) : 0u;
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (cls[t] == TILE_EMPTY) continue;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const f32x4 s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa0, kf[t], z, 0, 0, 0);
        const f32x4 s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa1, kf[t], z, 0, 0, 0);
        const f32x4 d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(oa0, vf[t], z, 0, 0, 0);
        const f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(oa1, vf[t], z, 0, 0, 0);
        float pr[8], ds[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int qi = q0 + slot32(g, e);
          float pe = fexp2((e < 4 ? s0[e] : s1[e - 4]) * sc2 - lq[e]);
          if (cls[t] == TILE_MIXED && !(qi < p.L && kj[t] < p.L && visible(qi, kj[t], p.x_len, xl, yl))) pe = 0.f;
          const float dm = p.drop_thr ? drop_mult2(p, drw[e], kj[t]) : 1.f;
          const float dp = (e < 4 ? d0[e] : d1[e - 4]);
          ds[e] = pe * (dp * dm - dq8[e]) * p.scale;
          pr[e] = pe * dm;
        }
        const bf16x8 pf = pack8(pr), dsf = pack8(ds);
        dvt[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dt0, pf, dvt[t][0], 0, 0, 0);
        dvt[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dt1, pf, dvt[t][1], 0, 0, 0);
        dkt[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt0, dsf, dkt[t][0], 0, 0, 0);
        dkt[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt1, dsf, dkt[t][1], 0, 0, 0);
      }
    }
  }
  bf16_t* dK = reinterpret_cast<bf16_t*>(p.dk) + b * p.sb + h * p.sh;
  bf16_t* dV = reinterpret_cast<bf16_t*>(p.dv) + b * p.sb + h * p.sh;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (kj[t] >= p.L) continue;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      bf16_t a4[4], b4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { a4[r] = f2bf(dkt[t][mt][r]); b4[r] = f2bf(dvt[t][mt][r]); }
      *reinterpret_cast<uint2*>(dK + kj[t] * p.sl + mt * 16 + g * 4) = *reinterpret_cast<uint2*>(a4);
      *reinterpret_cast<uint2*>(dV + kj[t] * p.sl + mt * 16 + g * 4) = *reinterpret_cast<uint2*>(b4);
    }
  }
}

// delta[b,h,q] = sum_d dO * O ; rows walked in memory order (b, q, h), 16-byte loads, 64/V lanes per row
template <typename T>
__global__ void attn_delta(AP p, int Dh) {
  constexpr int V = 16 / sizeof(T);
  const int lpr = Dh / V;                           // lanes per row (4 for bf16 D=32, 8 for f32 D=32)
  const long rows = (long)p.B * p.L * p.H;
  const T* O = reinterpret_cast<const T*>(p.o);
  const T* dO = reinterpret_cast<const T*>(p.d_o);
  float* delta = const_cast<float*>(p.delta);
  const long gid = blockIdx.x * (long)blockDim.x + threadIdx.x;
  const long row = gid / lpr;
  const int part = (int)(gid % lpr);
  float acc = 0.f;
  int q = 0, h = 0, b = 0;
  if (row < rows) {
    h = (int)(row % p.H);
    q = (int)((row / p.H) % p.L);
    b = (int)(row / ((long)p.H * p.L));
    const long off = b * p.ob + q * p.ol + h * p.oh + part * V;
    const uint4 a = *reinterpret_cast<const uint4*>(O + off);
    const uint4 c = *reinterpret_cast<const uint4*>(dO + off);
    const T* pa = reinterpret_cast<const T*>(&a);
    const T* pc = reinterpret_cast<const T*>(&c);
#pragma unroll
    for (int e = 0; e < V; ++e) acc += to_f<T>(pa[e]) * to_f<T>(pc[e]);
  }
  for (int o = lpr >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (row < rows && part == 0) delta[((long)b * p.H + h) * p.L + q] = acc;
}

// ---------------------------------------------------------------------------------------------------------
// fp32 parity path: one thread per query row (forward, dQ) / per key row (dK, dV); any head_dim <= 64
// ---------------------------------------------------------------------------------------------------------
template <int DH>
__global__ void attn_fwd_f32(AP p) {
  const long total = (long)p.B * p.H * p.L;
  const float* Q = reinterpret_cast<const float*>(p.q);
  const float* K = reinterpret_cast<const float*>(p.k);
  const float* V = reinterpret_cast<const float*>(p.v);
  float* O = reinterpret_cast<float*>(p.out);
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int q = (int)(i % p.L), h = (int)((i / p.L) % p.H), b = (int)(i / ((long)p.L * p.H));
  const int xl = p.x_lens[b], yl = p.y_lens[b];
  float qv[DH], o[DH];
  const float* qp = Q + b * p.sb + q * p.sl + h * p.sh;
#pragma unroll
  for (int d = 0; d < DH; ++d) { qv[d] = qp[d] * p.scale; o[d] = 0.f; }
  float m = -INFINITY, l = 0.f;
  const int kmax = q < p.x_len ? p.x_len : q + 1;
  for (int j = 0; j < kmax; ++j) {
    if (!visible(q, j, p.x_len, xl, yl)) continue;
    const float* kp = K + b * p.sb + j * p.sl + h * p.sh;
    const float* vp = V + b * p.sb + j * p.sl + h * p.sh;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < DH; ++d) s += qv[d] * kp[d];
    const float mn = fmaxf(m, s);
    const float alpha = expf(m - mn), pr = expf(s - mn);
    l = l * alpha + pr;
    const float prd = pr * drop_mult(p, (unsigned)(b * p.H + h), q, j);
#pragma unroll
    for (int d = 0; d < DH; ++d) o[d] = o[d] * alpha + prd * vp[d];
    m = mn;
  }
  const float inv = l > 0.f ? 1.f / l : 0.f;
  float* op = O + b * p.ob + q * p.ol + h * p.oh;
#pragma unroll
  for (int d = 0; d < DH; ++d) op[d] = o[d] * inv;
  p.lse[i] = m + logf(l);
}

template <int DH>
__global__ void attn_bwd_dq_f32(AP p) {
  const long total = (long)p.B * p.H * p.L;
  const float* Q = reinterpret_cast<const float*>(p.q);
  const float* K = reinterpret_cast<const float*>(p.k);
  const float* V = reinterpret_cast<const float*>(p.v);
  const float* dO = reinterpret_cast<const float*>(p.d_o);
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int q = (int)(i % p.L), h = (int)((i / p.L) % p.H), b = (int)(i / ((long)p.L * p.H));
  const int xl = p.x_lens[b], yl = p.y_lens[b];
  float qv[DH], dov[DH], dq[DH];
  const float* qp = Q + b * p.sb + q * p.sl + h * p.sh;
  const float* dp_ = dO + b * p.ob + q * p.ol + h * p.oh;
#pragma unroll
  for (int d = 0; d < DH; ++d) { qv[d] = qp[d]; dov[d] = dp_[d]; dq[d] = 0.f; }
  const float lse = p.lse[i], dl = p.delta[i];
  const int kmax = q < p.x_len ? p.x_len : q + 1;
  for (int j = 0; j < kmax; ++j) {
    if (!visible(q, j, p.x_len, xl, yl)) continue;
    const float* kp = K + b * p.sb + j * p.sl + h * p.sh;
    const float* vp = V + b * p.sb + j * p.sl + h * p.sh;
    float s = 0.f, dpv = 0.f;
#pragma unroll
    for (int d = 0; d < DH; ++d) { s += qv[d] * kp[d]; dpv += dov[d] * vp[d]; }
    const float pr = expf(s * p.scale - lse);
    const float ds = pr * (dpv * drop_mult(p, (unsigned)(b * p.H + h), q, j) - dl) * p.scale;
#pragma unroll
    for (int d = 0; d < DH; ++d) dq[d] += ds * kp[d];
  }
  float* dqp = reinterpret_cast<float*>(p.dq) + b * p.sb + q * p.sl + h * p.sh;
#pragma unroll
  for (int d = 0; d < DH; ++d) dqp[d] = dq[d];
}

template <int DH>
__global__ void attn_bwd_dkv_f32(AP p) {
  const long total = (long)p.B * p.H * p.L;
  const float* Q = reinterpret_cast<const float*>(p.q);
  const float* K = reinterpret_cast<const float*>(p.k);
  const float* V = reinterpret_cast<const float*>(p.v);
  const float* dO = reinterpret_cast<const float*>(p.d_o);
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int j = (int)(i % p.L), h = (int)((i / p.L) % p.H), b = (int)(i / ((long)p.L * p.H));
  const int xl = p.x_lens[b], yl = p.y_lens[b];
  float kv[DH], vv[DH], dk[DH], dv[DH];
  const float* kp = K + b * p.sb + j * p.sl + h * p.sh;
  const float* vp = V + b * p.sb + j * p.sl + h * p.sh;
#pragma unroll
  for (int d = 0; d < DH; ++d) { kv[d] = kp[d]; vv[d] = vp[d]; dk[d] = 0.f; dv[d] = 0.f; }
  const int q0 = j >= p.x_len ? j : 0;
  for (int q = q0; q < p.L; ++q) {
    if (!visible(q, j, p.x_len, xl, yl)) continue;
    const float* qp = Q + b * p.sb + q * p.sl + h * p.sh;
    const float* dp_ = dO + b * p.ob + q * p.ol + h * p.oh;
    const long li = ((long)b * p.H + h) * p.L + q;
    float s = 0.f, dpv = 0.f;
#pragma unroll
    for (int d = 0; d < DH; ++d) { s += qp[d] * kv[d]; dpv += dp_[d] * vv[d]; }
    const float pr = expf(s * p.scale - p.lse[li]);
    const float dm = drop_mult(p, (unsigned)(b * p.H + h), q, j);
    const float ds = pr * (dpv * dm - p.delta[li]) * p.scale;
#pragma unroll
    for (int d = 0; d < DH; ++d) { dv[d] += pr * dm * dp_[d]; dk[d] += ds * qp[d]; }
  }
  float* dkp = reinterpret_cast<float*>(p.dk) + b * p.sb + j * p.sl + h * p.sh;
  float* dvp = reinterpret_cast<float*>(p.dv) + b * p.sb + j * p.sl + h * p.sh;
#pragma unroll
  for (int d = 0; d < DH; ++d) { dkp[d] = dk[d]; dvp[d] = dv[d]; }
}

int check(const evt_attn_params* a) {
  if (!a || a->B <= 0 || a->L <= 0 || a->H <= 0) return EVT_EINVAL;
  if (a->dtype == EVT_DT_BF16) {
    if (a->D != 32) return EVT_ENOTSUP;
    if (a->q_stride_l % 8 || a->q_stride_h % 8 || a->q_stride_b % 8 || a->o_stride_l % 8 || a->o_stride_h % 8 ||
        a->o_stride_b % 8)
      return EVT_EINVAL;  // 16-byte row loads
  } else if (a->dtype == EVT_DT_F32) {
    if (a->D != 32 && a->D != 64) return EVT_ENOTSUP;
  } else return EVT_EINVAL;
  if (a->x_len < 0 || a->x_len > a->L) return EVT_EINVAL;
  if (a->dropout_p < 0.f || a->dropout_p >= 1.f) return EVT_EINVAL;
  return EVT_OK;
}

AP make_ap(const evt_attn_params* a) {
  AP p{};
  p.B = a->B; p.L = a->L; p.H = a->H; p.x_len = a->x_len;
  p.sb = a->q_stride_b; p.sl = a->q_stride_l; p.sh = a->q_stride_h;
  p.ob = a->o_stride_b; p.ol = a->o_stride_l; p.oh = a->o_stride_h;
  p.scale = 1.0f / sqrtf((float)a->D);
  if (a->dropout_p > 0.f) {
    const double thr = (double)a->dropout_p * 4294967296.0;
    p.drop_thr = thr >= 4294967295.0 ? 4294967295u : (unsigned)thr;
    p.keep_scale = 1.0f / (1.0f - a->dropout_p);
  } else { p.drop_thr = 0u; p.keep_scale = 1.f; }
  p.seed = a->seed;
  return p;
}

}  // namespace

extern "C" {

int evt_attn_prefixlm_fwd(const evt_attn_params* a, const void* q, const void* k, const void* v, const int32_t* x_lens,
                          const int32_t* y_lens, void* o, float* lse, void* stream) {
  int rc = check(a);
  if (rc) return rc;
  if (!q || !k || !v || !o || !lse || !x_lens || !y_lens) return EVT_EINVAL;
  AP p = make_ap(a);
  p.q = q; p.k = k; p.v = v; p.out = o; p.lse = lse; p.x_lens = x_lens; p.y_lens = y_lens;
  hipStream_t st = (hipStream_t)stream;
  if (a->dtype == EVT_DT_BF16) {
    hipLaunchKernelGGL(attn_fwd_bf16, dim3((a->L + 127) / 128, a->B * a->H), dim3(256), 0, st, p);
  } else {
    const long total = (long)a->B * a->H * a->L;
    const int blocks = (int)((total + 63) / 64);
    if (a->D == 32) hipLaunchKernelGGL(attn_fwd_f32<32>, dim3(blocks), dim3(64), 0, st, p);
    else hipLaunchKernelGGL(attn_fwd_f32<64>, dim3(blocks), dim3(64), 0, st, p);
  }
  return evt_check_launch();
}

int evt_attn_prefixlm_bwd(const evt_attn_params* a, const void* q, const void* k, const void* v, const void* o,
                          const void* d_o, const float* lse, const int32_t* x_lens, const int32_t* y_lens, void* dq,
                          void* dk, void* dv, float* delta_ws, void* stream) {
  int rc = check(a);
  if (rc) return rc;
  if (!q || !k || !v || !o || !d_o || !lse || !dq || !dk || !dv || !delta_ws || !x_lens || !y_lens) return EVT_EINVAL;
  AP p = make_ap(a);
  p.q = q; p.k = k; p.v = v; p.o = o; p.d_o = d_o; p.lse = const_cast<float*>(lse); p.delta = delta_ws;
  p.dq = dq; p.dk = dk; p.dv = dv; p.x_lens = x_lens; p.y_lens = y_lens;
  hipStream_t st = (hipStream_t)stream;
  const long total = (long)a->B * a->H * a->L;
  const int lpr = a->D / (a->dtype == EVT_DT_BF16 ? 8 : 4);
  const int dblocks = (int)((total * lpr + 255) / 256);
  if (a->dtype == EVT_DT_BF16) {
    hipLaunchKernelGGL(attn_delta<bf16_t>, dim3(dblocks), dim3(256), 0, st, p, a->D);
    hipLaunchKernelGGL(attn_bwd_dkv_bf16, dim3((a->L + 127) / 128, a->B * a->H), dim3(256), 0, st, p);
    hipLaunchKernelGGL(attn_bwd_dq_bf16, dim3((a->L + 127) / 128, a->B * a->H), dim3(256), 0, st, p);
  } else {
    hipLaunchKernelGGL(attn_delta<float>, dim3(dblocks), dim3(256), 0, st, p, a->D);
    const int blocks = (int)((total + 63) / 64);
    if (a->D == 32) {
      hipLaunchKernelGGL(attn_bwd_dkv_f32<32>, dim3(blocks), dim3(64), 0, st, p);
      hipLaunchKernelGGL(attn_bwd_dq_f32<32>, dim3(blocks), dim3(64), 0, st, p);
    } else {
      hipLaunchKernelGGL(attn_bwd_dkv_f32<64>, dim3(blocks), dim3(64), 0, st, p);
      hipLaunchKernelGGL(attn_bwd_dq_f32<64>, dim3(blocks), dim3(64), 0, st, p);
    }
  }
  return evt_check_launch();
}

}  // extern "C"
