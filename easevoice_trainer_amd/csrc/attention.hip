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
#include <type_traits>
#include <cstdlib>

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
  unsigned drop_thr;   // thr16 << 16: a 16-bit hash field f is kept iff (f << 16) >= drop_thr ; 0 = no dropout
  unsigned seed;
  float keep_scale;    // 1 / (1 - p)
};

// Counter-based dropout mask (attention dropout of SDPA, patched_mha_with_cache.py:452-454): a pure function of
// (seed, b*H+h, query, key) so forward and both backward kernels regenerate the same mask with no storage.
// ONE 32-bit hash decides TWO scores: keys k and k ^ 16 of a query share the hash of kp = k & ~16, key kp is kept iff
// the low 16 bits >= thr16, key kp + 16 iff the high 16 bits >= thr16 (thr16 = round(p * 65536): p = 0.1 -> 0.100006;
// keep_scale = 1 / (1 - thr16 / 65536) keeps the expectation exact).  The two keys are the e / e + 4 score slots of one
// lane in the S^T tiles of forward / dQ and the two key tiles of one lane in dK/dV, so the per-score cost is half a
// hash (an add, an xor-shift, a 24-bit multiply-xor: full-rate VALU only) plus one compare and one select.
// The per-query part is a full avalanche hash (computed once per query and kernel, not per score).
constexpr unsigned DROP_KC = 0x85EBCBu;
__device__ __forceinline__ unsigned drop_row(const AP& p, unsigned bh, int qi) {
  unsigned a = ((bh << 11) ^ (unsigned)qi) + p.seed * 0x9E3779B1u;
  a ^= a >> 16; a *= 0x85EBCA6Bu; a ^= a >> 13; a *= 0xC2B2AE35u; a ^= a >> 16;
  return a;
}
__device__ __forceinline__ unsigned drop_pair(unsigned x) {      // x = drop_row + umul24(kp, DROP_KC)
  x ^= x >> 15;
  return __umul24(x, 0x2C1B3Du) ^ (x >> 9);
}
__device__ __forceinline__ bool keep_lo(unsigned h, unsigned thr_hi) { return (h << 16) >= thr_hi; }
__device__ __forceinline__ bool keep_hi(unsigned h, unsigned thr_hi) { return h >= thr_hi; }
__device__ __forceinline__ float drop_mult(const AP& p, unsigned bh, int qi, int kj) {
  if (p.drop_thr == 0u) return 1.f;
  const unsigned h = drop_pair(drop_row(p, bh, qi) + __umul24((unsigned)(kj & ~16), DROP_KC));
  return ((kj & 16) ? keep_hi(h, p.drop_thr) : keep_lo(h, p.drop_thr)) ? p.keep_scale : 0.f;
}

__device__ __forceinline__ bool visible(int qi, int kj, int x_len, int xl, int yl) {
  const bool key_ok = kj < x_len ? (kj < xl) : (kj - x_len < yl);
  if (!key_ok) return false;
  return qi < x_len ? (kj < x_len) : (kj <= qi);
}

// two ds_read_b64_tr_b16 (transposing 16-bit LDS reads) -> one MFMA A fragment.  The builtin (not inline asm) lets the
// compiler track the LDS counter, so all fragment reads of a key sub-block are in flight behind ONE wait.
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ h16x8 tr2(const h16_t* p0, const h16_t* p1) {
  typedef __attribute__((address_space(3))) s16x4* lds_p;
  union { struct { s16x4 lo, hi; } h; h16x8 v; } r;
  r.h.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p0));
  r.h.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p1));
  return r.v;
}

__device__ __forceinline__ h16x8 pack8(const float* p) {
  union { h16x8 v; h16_t e[8]; } r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.e[i] = f2h(p[i]);
  return r.v;
}

__device__ __forceinline__ h16x8 ld8(const h16_t* p) { return *reinterpret_cast<const h16x8*>(p); }

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
// max(a, b) as med3(a, b, +inf): fmaxf() makes the compiler quiet possible signalling NaNs first (a v_max_f32 x, x, x
// per operand that comes out of an MFMA -- 12 extra VALU instructions per 16 x 32 score tile); the med3 intrinsic is
// emitted as is.  Scores are finite or -inf here, never NaN.  (Not an option: inline-asm v_max3 -- the hazard recogniser
// does not treat an asm statement as a VALU reader of a just-issued MFMA result; -mno-amdgpu-ieee -- device-library
// functions, blockIdx included, are then no longer inlined.)
__device__ __forceinline__ float vmax(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, INFINITY); }
__device__ __forceinline__ float vmax3(float a, float b, float c) { return vmax(vmax(a, b), c); }
__device__ __forceinline__ float max8(const float* s) {
  return vmax(vmax3(s[0], s[1], s[2]), vmax3(vmax3(s[3], s[4], s[5]), s[6], s[7]));
}
__device__ __forceinline__ float quad_max(float v) {
  unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  v = vmax(__uint_as_float(r[0]), __uint_as_float(r[1]));
  u = __float_as_uint(v);
  auto r2 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return vmax(__uint_as_float(r2[0]), __uint_as_float(r2[1]));
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
// Shared structure of the three bf16 kernels: a block owns 128 rows of one (b, h) (4 waves x 2 tiles of 16) and walks
// the other axis in blocks of 64 rows through a DOUBLE-BUFFERED LDS tile pair: the next block's global loads are issued
// before the current block is multiplied and land in registers under it; they are written to the other buffer after
// the multiply, ONE barrier per block.  Scale folding: scores stay raw, exp2(fma(s, scale*log2e, -m*scale*log2e));
// the 1/(1-p) of dropout and the 1/sqrt(d) of dS are applied once to the accumulators at the end.
// ---------------------------------------------------------------------------------------------------------
// register staging of one 64-row tile pair (256 threads x 16 bytes per tensor); plain locals so they stay in VGPRs
#define EVT_TILE_REGS uint4 tl_a = make_uint4(0, 0, 0, 0), tl_b = make_uint4(0, 0, 0, 0); const int tl_r = tid >> 2, tl_c8 = tid & 3
#define EVT_TILE_LOAD(A, sa, B, sb_, row0)                                              \
  {                                                                                     \
    const int tl_j = min((row0) + tl_r, p.L - 1);                                       \
    tl_a = *reinterpret_cast<const uint4*>((A) + tl_j * (sa) + tl_c8 * 8);             \
    tl_b = *reinterpret_cast<const uint4*>((B) + tl_j * (sb_) + tl_c8 * 8);            \
  }
#define EVT_TILE_STORE(As, Bs)                                                          \
  {                                                                                     \
    *reinterpret_cast<uint4*>((As) + tl_r * PITCH + tl_c8 * 8) = tl_a;                  \
    *reinterpret_cast<uint4*>((Bs) + tl_r * PITCH + tl_c8 * 8) = tl_b;                  \
  }

// ---------------------------------------------------------------------------------------------------------
// forward (bf16): block = 4 waves x 2 query tiles (128 queries) of one (b, h); key blocks of 64 through LDS
// ---------------------------------------------------------------------------------------------------------
template <bool JOINT>
__global__ __launch_bounds__(256, 2) void attn_fwd_bf16(AP p) {
  __shared__ __attribute__((aligned(16))) h16_t Ks[2][64 * PITCH];
  __shared__ __attribute__((aligned(16))) h16_t Vs[2][64 * PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  // 1-D grid, LONGEST BLOCKS FIRST: query block nqb-1 (walks every key) of all (b, h), then nqb-2, ... -- the short text
  // blocks fill the tail (with the (b, h)-major order the last dispatched 16-iteration blocks ran on an empty chip)
  const int BH = p.B * p.H;
  const int bh = blockIdx.x % BH;
  const int b = bh / p.H, h = bh % p.H;
  const int qblk = ((p.L + 127) / 128 - 1 - blockIdx.x / BH) * 128;
  const int xl = p.x_lens[b], yl = p.y_lens[b];
  const h16_t* Q = reinterpret_cast<const h16_t*>(p.q) + b * p.sb + h * p.sh;
  const h16_t* K = reinterpret_cast<const h16_t*>(p.k) + b * p.sb + h * p.sh;
  const h16_t* V = reinterpret_cast<const h16_t*>(p.v) + b * p.sb + h * p.sh;
  const float sc2 = p.scale * LOG2E;
  const unsigned thr = p.drop_thr;

  h16x8 qf[2];
  f32x4 ot[2][2];
  float m[2], ms[2], l[2];          // running maximum (raw scores), the same times scale*log2(e), running sum
  int qi[2], qt0[2];
  unsigned rowc[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    qt0[t] = qblk + wave * 32 + t * 16;
    qi[t] = qt0[t] + n;
    const int qc = min(qi[t], p.L - 1);
    qf[t] = ld8(Q + qc * p.sl + g * 8);
    ot[t][0] = ot[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    m[t] = -INFINITY;
    ms[t] = 0.f;
    l[t] = 0.f;
    const unsigned row = thr ? drop_row(p, (unsigned)bh, qi[t]) : 0u;
#pragma unroll
    for (int e = 0; e < 4; ++e) rowc[t][e] = row + __umul24((unsigned)(g * 4 + e), DROP_KC);
  }
  const int qlast = min(qblk + 127, p.L - 1);
  const int kmax = (qlast < p.x_len) ? p.x_len : qlast + 1;
  EVT_TILE_REGS;
  EVT_TILE_LOAD(K, p.sl, V, p.sl, 0);
  EVT_TILE_STORE(Ks[0], Vs[0]);
  __syncthreads();
  int buf = 0;
  for (int kb = 0; kb < kmax; kb += 64, buf ^= 1) {
    const bool more = kb + 64 < kmax;
    if (more) EVT_TILE_LOAD(K, p.sl, V, p.sl, kb + 64);
    const h16_t* Kc = Ks[buf];
    const h16_t* Vc = Vs[buf];
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
      const int k0 = kb + sb * 32;
      int cls[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) cls[t] = classify(qt0[t], qt0[t] + 15, k0, k0 + 31, p.L, p.x_len, xl, yl);
      if (cls[0] == TILE_EMPTY && cls[1] == TILE_EMPTY) continue;
      const h16x8 ka0 = ld8(Kc + (sb * 32 + n) * PITCH + g * 8);
      const h16x8 ka1 = ld8(Kc + (sb * 32 + 16 + n) * PITCH + g * 8);
      const h16_t* vrow = Vc + (sb * 32 + g * 4 + (n >> 2)) * PITCH + 4 * (n & 3);
      const h16x8 va0 = tr2(vrow, vrow + 16 * PITCH);
      const h16x8 va1 = tr2(vrow + 16, vrow + 16 * PITCH + 16);
      const unsigned k0c = __umul24((unsigned)k0, DROP_KC);
      // both query tiles in ONE straight-line body (two independent dependency chains for the scheduler); the masked
      // variant (diagonal / padding tiles, a few percent of the work) evaluates visible() per score in both tiles --
      // a fully masked tile leaves every exp2 at 0 and the running statistics untouched
      auto body = [&](auto masked_tag, auto t0_tag, auto t1_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        constexpr int T0 = decltype(t0_tag)::value, T1 = decltype(t1_tag)::value;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        f32x4 sa[2], sb2[2];
#pragma unroll
        for (int t = T0; t < T1; ++t) {
          sa[t] = EVT_MFMA_16x16x32(ka0, qf[t], z, 0, 0, 0);
          sb2[t] = EVT_MFMA_16x16x32(ka1, qf[t], z, 0, 0, 0);
        }
        float s[2][8];
#pragma unroll
        for (int t = T0; t < T1; ++t) {
#pragma unroll
          for (int e = 0; e < 8; ++e) s[t][e] = e < 4 ? sa[t][e] : sb2[t][e - 4];
          if (MASKED) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const int kj = k0 + slot32(g, e);
              if (!(kj < p.L && visible(qi[t], kj, p.x_len, xl, yl))) s[t][e] = -INFINITY;
            }
          }
          const float mx = quad_max(max8(s[t]));
          // the accumulators are rescaled only when some query of the tile saw a new maximum (wave-uniform branch)
          if (__builtin_amdgcn_ballot_w64(mx > m[t]) != 0ull) {
            const float mn = vmax(m[t], mx);
            const float mns = mn == -INFINITY ? 0.f : mn * sc2;   // nothing visible yet: every exp2 below gives 0
            const float alpha = fexp2(fmaf(m[t], sc2, -mns));
            l[t] *= alpha;
#pragma unroll
            for (int r = 0; r < 4; ++r) { ot[t][0][r] *= alpha; ot[t][1][r] *= alpha; }
            m[t] = mn;
            ms[t] = mns;
          }
          float sum = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) { s[t][e] = fexp2(fmaf(s[t][e], sc2, -ms[t])); sum += s[t][e]; }
          l[t] += quad_sum(sum);
          if (thr) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const unsigned hh = drop_pair(rowc[t][e] + k0c);
              s[t][e] = keep_lo(hh, thr) ? s[t][e] : 0.f;
              s[t][e + 4] = keep_hi(hh, thr) ? s[t][e + 4] : 0.f;
            }
          }
        }
#pragma unroll
        for (int t = T0; t < T1; ++t) {
          const h16x8 pf = pack8(s[t]);
          ot[t][0] = EVT_MFMA_16x16x32(va0, pf, ot[t][0], 0, 0, 0);
          ot[t][1] = EVT_MFMA_16x16x32(va1, pf, ot[t][1], 0, 0, 0);
        }
      };
      using I0 = std::integral_constant<int, 0>;
      using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>;
      if (JOINT) {
        if (cls[0] == TILE_FULL && cls[1] == TILE_FULL) body(std::false_type{}, I0{}, I2{});
        else body(std::true_type{}, I0{}, I2{});
      } else {
        if (cls[0] == TILE_FULL) body(std::false_type{}, I0{}, I1{});
        else if (cls[0] == TILE_MIXED) body(std::true_type{}, I0{}, I1{});
        if (cls[1] == TILE_FULL) body(std::false_type{}, I1{}, I2{});
        else if (cls[1] == TILE_MIXED) body(std::true_type{}, I1{}, I2{});
      }
    }
    if (more) EVT_TILE_STORE(Ks[buf ^ 1], Vs[buf ^ 1]);
    __syncthreads();
  }
  h16_t* O = reinterpret_cast<h16_t*>(p.out) + b * p.ob + h * p.oh;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (qi[t] >= p.L) continue;
    const float inv = l[t] > 0.f ? p.keep_scale / l[t] : 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      h16_t o4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) o4[r] = f2h(ot[t][mt][r] * inv);
      *reinterpret_cast<uint2*>(O + qi[t] * p.ol + mt * 16 + g * 4) = *reinterpret_cast<uint2*>(o4);
    }
    if (g == 0) p.lse[((long)b * p.H + h) * p.L + qi[t]] = (m[t] * sc2 + __log2f(l[t])) * LN2;
  }
}

// ---------------------------------------------------------------------------------------------------------
// dQ (bf16): same ownership as forward; dQ^T += K^T dS^T
// dS = keep_scale * P o (M o dP_drop - delta / keep_scale): the constant factors go to the epilogue
// ---------------------------------------------------------------------------------------------------------
template <bool JOINT>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_bf16(AP p) {
  __shared__ __attribute__((aligned(16))) h16_t Ks[2][64 * PITCH];
  __shared__ __attribute__((aligned(16))) h16_t Vs[2][64 * PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  // 1-D grid, LONGEST BLOCKS FIRST: query block nqb-1 (walks every key) of all (b, h), then nqb-2, ... -- the short text
  // blocks fill the tail (with the (b, h)-major order the last dispatched 16-iteration blocks ran on an empty chip)
  const int BH = p.B * p.H;
  const int bh = blockIdx.x % BH;
  const int b = bh / p.H, h = bh % p.H;
  const int qblk = ((p.L + 127) / 128 - 1 - blockIdx.x / BH) * 128;
  const int xl = p.x_lens[b], yl = p.y_lens[b];
  const h16_t* Q = reinterpret_cast<const h16_t*>(p.q) + b * p.sb + h * p.sh;
  const h16_t* K = reinterpret_cast<const h16_t*>(p.k) + b * p.sb + h * p.sh;
  const h16_t* V = reinterpret_cast<const h16_t*>(p.v) + b * p.sb + h * p.sh;
  const h16_t* dO = reinterpret_cast<const h16_t*>(p.d_o) + b * p.ob + h * p.oh;
  const float sc2 = p.scale * LOG2E;
  const unsigned thr = p.drop_thr;
  const float inv_ks = 1.f / p.keep_scale;
  h16x8 qf[2], dof[2];
  f32x4 dqt[2][2];
  float lse2[2], dlk[2];
  int qi[2], qt0[2];
  unsigned rowc[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    qt0[t] = qblk + wave * 32 + t * 16;
    qi[t] = qt0[t] + n;
    const int qc = min(qi[t], p.L - 1);
    qf[t] = ld8(Q + qc * p.sl + g * 8);
    dof[t] = ld8(dO + qc * p.ol + g * 8);
    lse2[t] = p.lse[((long)b * p.H + h) * p.L + qc] * LOG2E;
    dlk[t] = p.delta[((long)b * p.H + h) * p.L + qc] * inv_ks;
    dqt[t][0] = dqt[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned row = thr ? drop_row(p, (unsigned)bh, qi[t]) : 0u;
#pragma unroll
    for (int e = 0; e < 4; ++e) rowc[t][e] = row + __umul24((unsigned)(g * 4 + e), DROP_KC);
  }
  const int qlast = min(qblk + 127, p.L - 1);
  const int kmax = (qlast < p.x_len) ? p.x_len : qlast + 1;
  EVT_TILE_REGS;
  EVT_TILE_LOAD(K, p.sl, V, p.sl, 0);
  EVT_TILE_STORE(Ks[0], Vs[0]);
  __syncthreads();
  int buf = 0;
  for (int kb = 0; kb < kmax; kb += 64, buf ^= 1) {
    const bool more = kb + 64 < kmax;
    if (more) EVT_TILE_LOAD(K, p.sl, V, p.sl, kb + 64);
    const h16_t* Kc = Ks[buf];
    const h16_t* Vc = Vs[buf];
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
      const int k0 = kb + sb * 32;
      int cls[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) cls[t] = classify(qt0[t], qt0[t] + 15, k0, k0 + 31, p.L, p.x_len, xl, yl);
      if (cls[0] == TILE_EMPTY && cls[1] == TILE_EMPTY) continue;
      const h16x8 ka0 = ld8(Kc + (sb * 32 + n) * PITCH + g * 8);
      const h16x8 ka1 = ld8(Kc + (sb * 32 + 16 + n) * PITCH + g * 8);
      const h16x8 va0 = ld8(Vc + (sb * 32 + n) * PITCH + g * 8);
      const h16x8 va1 = ld8(Vc + (sb * 32 + 16 + n) * PITCH + g * 8);
      const h16_t* krow = Kc + (sb * 32 + g * 4 + (n >> 2)) * PITCH + 4 * (n & 3);
      const h16x8 kt0 = tr2(krow, krow + 16 * PITCH);
      const h16x8 kt1 = tr2(krow + 16, krow + 16 * PITCH + 16);
      const unsigned k0c = __umul24((unsigned)k0, DROP_KC);
      auto body = [&](auto masked_tag, auto t0_tag, auto t1_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        constexpr int T0 = decltype(t0_tag)::value, T1 = decltype(t1_tag)::value;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        f32x4 s0[2], s1[2], d0[2], d1[2];
#pragma unroll
        for (int t = T0; t < T1; ++t) {
          s0[t] = EVT_MFMA_16x16x32(ka0, qf[t], z, 0, 0, 0);
          s1[t] = EVT_MFMA_16x16x32(ka1, qf[t], z, 0, 0, 0);
          d0[t] = EVT_MFMA_16x16x32(va0, dof[t], z, 0, 0, 0);
          d1[t] = EVT_MFMA_16x16x32(va1, dof[t], z, 0, 0, 0);
        }
        float ds[2][8];
#pragma unroll
        for (int t = T0; t < T1; ++t) {
          float pr[8], dp[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            pr[e] = fexp2(fmaf(e < 4 ? s0[t][e] : s1[t][e - 4], sc2, -lse2[t]));
            dp[e] = e < 4 ? d0[t][e] : d1[t][e - 4];
          }
          if (MASKED) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const int kj = k0 + slot32(g, e);
              if (!(kj < p.L && visible(qi[t], kj, p.x_len, xl, yl))) pr[e] = 0.f;
            }
          }
          if (thr) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const unsigned hh = drop_pair(rowc[t][e] + k0c);
              dp[e] = keep_lo(hh, thr) ? dp[e] : 0.f;
              dp[e + 4] = keep_hi(hh, thr) ? dp[e + 4] : 0.f;
            }
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) ds[t][e] = pr[e] * (dp[e] - dlk[t]);
        }
#pragma unroll
        for (int t = T0; t < T1; ++t) {
          const h16x8 dsf = pack8(ds[t]);
          dqt[t][0] = EVT_MFMA_16x16x32(kt0, dsf, dqt[t][0], 0, 0, 0);
          dqt[t][1] = EVT_MFMA_16x16x32(kt1, dsf, dqt[t][1], 0, 0, 0);
        }
      };
      using I0 = std::integral_constant<int, 0>;
      using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>;
      if (JOINT) {
        if (cls[0] == TILE_FULL && cls[1] == TILE_FULL) body(std::false_type{}, I0{}, I2{});
        else body(std::true_type{}, I0{}, I2{});
      } else {
        if (cls[0] == TILE_FULL) body(std::false_type{}, I0{}, I1{});
        else if (cls[0] == TILE_MIXED) body(std::true_type{}, I0{}, I1{});
        if (cls[1] == TILE_FULL) body(std::false_type{}, I1{}, I2{});
        else if (cls[1] == TILE_MIXED) body(std::true_type{}, I1{}, I2{});
      }
    }
    if (more) EVT_TILE_STORE(Ks[buf ^ 1], Vs[buf ^ 1]);
    __syncthreads();
  }
  h16_t* dQ = reinterpret_cast<h16_t*>(p.dq) + b * p.sb + h * p.sh;
  const float fin = p.scale * p.keep_scale;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (qi[t] >= p.L) continue;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      h16_t o4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) o4[r] = f2h(dqt[t][mt][r] * fin);
      *reinterpret_cast<uint2*>(dQ + qi[t] * p.sl + mt * 16 + g * 4) = *reinterpret_cast<uint2*>(o4);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// dK/dV (bf16): block = 4 waves x 2 key tiles (128 keys); query blocks of 64 through LDS.  A lane's two key tiles are
// the keys kp and kp + 16 of one dropout hash; the per-query hash part comes from an LDS table filled per query block.
// ---------------------------------------------------------------------------------------------------------
// one-tile variant: 3 waves per SIMD (168 registers, four values spilled outside the tile loop) -- measured 277 -> 233 us;
// the same squeeze on forward / dQ (5 waves, 96 registers) spills inside the loop and loses 30 % / 145 %
template <bool JOINT>
__global__ __launch_bounds__(256, JOINT ? 2 : 3) void attn_bwd_dkv_bf16(AP p) {
  __shared__ __attribute__((aligned(16))) h16_t Qs[2][64 * PITCH];
  __shared__ __attribute__((aligned(16))) h16_t Os[2][64 * PITCH];
  __shared__ __attribute__((aligned(16))) float lse_s[2][64];
  __shared__ __attribute__((aligned(16))) float dl_s[2][64];
  __shared__ __attribute__((aligned(16))) unsigned row_s[2][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  // 1-D grid, longest blocks first: key block 0 (seen by every query) of all (b, h), then 1, ...
  const int BH = p.B * p.H;
  const int bh = blockIdx.x % BH;
  const int b = bh / p.H, h = bh % p.H;
  const int kblk = (blockIdx.x / BH) * 128;
  const int xl = p.x_lens[b], yl = p.y_lens[b];
  const h16_t* Q = reinterpret_cast<const h16_t*>(p.q) + b * p.sb + h * p.sh;
  const h16_t* K = reinterpret_cast<const h16_t*>(p.k) + b * p.sb + h * p.sh;
  const h16_t* V = reinterpret_cast<const h16_t*>(p.v) + b * p.sb + h * p.sh;
  const h16_t* dO = reinterpret_cast<const h16_t*>(p.d_o) + b * p.ob + h * p.oh;
  const float sc2 = p.scale * LOG2E;
  const unsigned thr = p.drop_thr;
  const float inv_ks = 1.f / p.keep_scale;
  h16x8 kf[2], vf[2];
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
  const unsigned kpc = __umul24((unsigned)kj[0], DROP_KC);        // kj[0] has bit 4 clear: it is the pair's kp
  const int q_begin = (kblk >= p.x_len) ? (kblk / 64) * 64 : 0;
  EVT_TILE_REGS;
  float lse_r = 0.f, dl_r = 0.f;
  unsigned row_r = 0u;
  auto load_rows = [&](int qb) {
    if (tid < 64) {
      const int q2 = min(qb + tid, p.L - 1);
      lse_r = p.lse[((long)b * p.H + h) * p.L + q2] * LOG2E;
      dl_r = p.delta[((long)b * p.H + h) * p.L + q2] * inv_ks;
      row_r = thr ? drop_row(p, (unsigned)bh, qb + tid) : 0u;
    }
  };
  auto store_rows = [&](int bi) {
    if (tid < 64) { lse_s[bi][tid] = lse_r; dl_s[bi][tid] = dl_r; row_s[bi][tid] = row_r; }
  };
  EVT_TILE_LOAD(Q, p.sl, dO, p.ol, q_begin);
  load_rows(q_begin);
  EVT_TILE_STORE(Qs[0], Os[0]);
  store_rows(0);
  __syncthreads();
  int buf = 0;
  for (int qb = q_begin; qb < p.L; qb += 64, buf ^= 1) {
    const bool more = qb + 64 < p.L;
    if (more) { EVT_TILE_LOAD(Q, p.sl, dO, p.ol, qb + 64); load_rows(qb + 64); }
    const h16_t* Qc = Qs[buf];
    const h16_t* Oc = Os[buf];
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
      const int q0 = qb + sb * 32;
      int cls[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) cls[t] = classify(q0, q0 + 31, kt0[t], kt0[t] + 15, p.L, p.x_len, xl, yl);
      if (cls[0] == TILE_EMPTY && cls[1] == TILE_EMPTY) continue;
      const h16x8 qa0 = ld8(Qc + (sb * 32 + n) * PITCH + g * 8);
      const h16x8 qa1 = ld8(Qc + (sb * 32 + 16 + n) * PITCH + g * 8);
      const h16x8 oa0 = ld8(Oc + (sb * 32 + n) * PITCH + g * 8);
      const h16x8 oa1 = ld8(Oc + (sb * 32 + 16 + n) * PITCH + g * 8);
      const h16_t* qrow = Qc + (sb * 32 + g * 4 + (n >> 2)) * PITCH + 4 * (n & 3);
      const h16_t* orow = Oc + (sb * 32 + g * 4 + (n >> 2)) * PITCH + 4 * (n & 3);
      const h16x8 qt0 = tr2(qrow, qrow + 16 * PITCH), qt1 = tr2(qrow + 16, qrow + 16 * PITCH + 16);
      const h16x8 dt0 = tr2(orow, orow + 16 * PITCH), dt1 = tr2(orow + 16, orow + 16 * PITCH + 16);
      float lq[8], dq8[8];
      unsigned hh[8];
      {
        const f32x4 la = *reinterpret_cast<const f32x4*>(&lse_s[buf][sb * 32 + g * 4]);
        const f32x4 lb = *reinterpret_cast<const f32x4*>(&lse_s[buf][sb * 32 + 16 + g * 4]);
        const f32x4 da = *reinterpret_cast<const f32x4*>(&dl_s[buf][sb * 32 + g * 4]);
        const f32x4 db = *reinterpret_cast<const f32x4*>(&dl_s[buf][sb * 32 + 16 + g * 4]);
        const uint4 ra = *reinterpret_cast<const uint4*>(&row_s[buf][sb * 32 + g * 4]);
        const uint4 rb = *reinterpret_cast<const uint4*>(&row_s[buf][sb * 32 + 16 + g * 4]);
        const unsigned rr[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          lq[e] = e < 4 ? la[e] : lb[e - 4];
          dq8[e] = e < 4 ? da[e] : db[e - 4];
          hh[e] = thr ? drop_pair(rr[e] + kpc) : 0u;
        }
      }
      auto body = [&](auto masked_tag, auto t0_tag, auto t1_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        constexpr int T0 = decltype(t0_tag)::value, T1 = decltype(t1_tag)::value;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        f32x4 s0[2], s1[2], d0[2], d1[2];
#pragma unroll
        for (int t = T0; t < T1; ++t) {
          s0[t] = EVT_MFMA_16x16x32(qa0, kf[t], z, 0, 0, 0);
          s1[t] = EVT_MFMA_16x16x32(qa1, kf[t], z, 0, 0, 0);
          d0[t] = EVT_MFMA_16x16x32(oa0, vf[t], z, 0, 0, 0);
          d1[t] = EVT_MFMA_16x16x32(oa1, vf[t], z, 0, 0, 0);
        }
        float pr[2][8], ds[2][8];
#pragma unroll
        for (int t = T0; t < T1; ++t) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float pe = fexp2(fmaf(e < 4 ? s0[t][e] : s1[t][e - 4], sc2, -lq[e]));
            if (MASKED) {
              const int qi = q0 + slot32(g, e);
              if (!(qi < p.L && kj[t] < p.L && visible(qi, kj[t], p.x_len, xl, yl))) pe = 0.f;
            }
            float dp = e < 4 ? d0[t][e] : d1[t][e - 4];
            float pk = pe;
            if (thr) {
              const bool keep = t == 0 ? keep_lo(hh[e], thr) : keep_hi(hh[e], thr);
              dp = keep ? dp : 0.f;
              pk = keep ? pe : 0.f;
            }
            ds[t][e] = pe * (dp - dq8[e]);
            pr[t][e] = pk;
          }
        }
#pragma unroll
        for (int t = T0; t < T1; ++t) {
          const h16x8 pf = pack8(pr[t]), dsf = pack8(ds[t]);
          dvt[t][0] = EVT_MFMA_16x16x32(dt0, pf, dvt[t][0], 0, 0, 0);
          dvt[t][1] = EVT_MFMA_16x16x32(dt1, pf, dvt[t][1], 0, 0, 0);
          dkt[t][0] = EVT_MFMA_16x16x32(qt0, dsf, dkt[t][0], 0, 0, 0);
          dkt[t][1] = EVT_MFMA_16x16x32(qt1, dsf, dkt[t][1], 0, 0, 0);
        }
      };
      using I0 = std::integral_constant<int, 0>;
      using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>;
      if (JOINT) {
        if (cls[0] == TILE_FULL && cls[1] == TILE_FULL) body(std::false_type{}, I0{}, I2{});
        else body(std::true_type{}, I0{}, I2{});
      } else {
        if (cls[0] == TILE_FULL) body(std::false_type{}, I0{}, I1{});
        else if (cls[0] == TILE_MIXED) body(std::true_type{}, I0{}, I1{});
        if (cls[1] == TILE_FULL) body(std::false_type{}, I1{}, I2{});
        else if (cls[1] == TILE_MIXED) body(std::true_type{}, I1{}, I2{});
      }
    }
    if (more) { EVT_TILE_STORE(Qs[buf ^ 1], Os[buf ^ 1]); store_rows(buf ^ 1); }
    __syncthreads();
  }
  h16_t* dK = reinterpret_cast<h16_t*>(p.dk) + b * p.sb + h * p.sh;
  h16_t* dV = reinterpret_cast<h16_t*>(p.dv) + b * p.sb + h * p.sh;
  const float fk = p.scale * p.keep_scale, fv = p.keep_scale;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (kj[t] >= p.L) continue;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      h16_t a4[4], b4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { a4[r] = f2h(dkt[t][mt][r] * fk); b4[r] = f2h(dvt[t][mt][r] * fv); }
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

// two-tile bodies (more instruction-level parallelism, more registers) or one tile at a time: measurement switch
int g_attn_joint = getenv("EVT_ATTN_JOINT") ? atoi(getenv("EVT_ATTN_JOINT")) : 0;   // measured: one tile at a time is 4-9 % faster

int check(const evt_attn_params* a) {
  if (!a || a->B <= 0 || a->L <= 0 || a->H <= 0) return EVT_EINVAL;
  if (a->dtype == EVT_DT_HALF) {
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
    unsigned thr16 = (unsigned)((double)a->dropout_p * 65536.0 + 0.5);
    if (thr16 < 1u) thr16 = 1u;
    if (thr16 > 65535u) thr16 = 65535u;
    p.drop_thr = thr16 << 16;
    p.keep_scale = 65536.0f / (float)(65536u - thr16);
  } else { p.drop_thr = 0u; p.keep_scale = 1.f; }
  p.seed = a->seed;
  return p;
}

}  // namespace

extern "C" {

void evt_debug_attn_variant(int joint) { g_attn_joint = joint; }

int evt_attn_prefixlm_fwd(const evt_attn_params* a, const void* q, const void* k, const void* v, const int32_t* x_lens,
                          const int32_t* y_lens, void* o, float* lse, void* stream) {
  int rc = check(a);
  if (rc) return rc;
  if (!q || !k || !v || !o || !lse || !x_lens || !y_lens) return EVT_EINVAL;
  AP p = make_ap(a);
  p.q = q; p.k = k; p.v = v; p.out = o; p.lse = lse; p.x_lens = x_lens; p.y_lens = y_lens;
  hipStream_t st = (hipStream_t)stream;
  if (a->dtype == EVT_DT_HALF) {
    if (g_attn_joint) hipLaunchKernelGGL(attn_fwd_bf16<true>, dim3(((a->L + 127) / 128) * a->B * a->H), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(attn_fwd_bf16<false>, dim3(((a->L + 127) / 128) * a->B * a->H), dim3(256), 0, st, p);
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
  const int lpr = a->D / (a->dtype == EVT_DT_HALF ? 8 : 4);
  const int dblocks = (int)((total * lpr + 255) / 256);
  if (a->dtype == EVT_DT_HALF) {
    hipLaunchKernelGGL(attn_delta<h16_t>, dim3(dblocks), dim3(256), 0, st, p, a->D);
    const dim3 grid(((a->L + 127) / 128) * a->B * a->H);
    if (g_attn_joint) {
      hipLaunchKernelGGL(attn_bwd_dkv_bf16<true>, grid, dim3(256), 0, st, p);
      hipLaunchKernelGGL(attn_bwd_dq_bf16<true>, grid, dim3(256), 0, st, p);
    } else {
      hipLaunchKernelGGL(attn_bwd_dkv_bf16<false>, grid, dim3(256), 0, st, p);
      hipLaunchKernelGGL(attn_bwd_dq_bf16<false>, grid, dim3(256), 0, st, p);
    }
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
