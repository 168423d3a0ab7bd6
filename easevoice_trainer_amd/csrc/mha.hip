// Fused multi-head attention core of the s2 encoders, gfx950: the windowed relative-position self-attention of enc_p
// (attentions.py:214-292, window_size = 4), the window-less cross-attention of MRTE (mrte_model.py:25-61: queries = ssl
// frames, keys / values = phonemes) and the style encoder's self-attention (modules.py:605-682) are ONE kernel family:
//   scores[i][j] = scale * (q_i . k_j + [REL and |j-i| <= w] q_i . Ek[j-i+w]);  keys j >= lens_k[b] excluded
//   p = drop(softmax_j(scores));   out_i = sum_j p[i][j] (v_j + [REL and |j-i| <= w] Ev[j-i+w])
// which the reference runs as ~18 launches forward and ~40 backward per layer over [B, H, Tq, Tk] tensors (two bmm, two
// skew pad/reshape chains, masked_fill, softmax, dropout, two more bmm ...).  Here: one forward launch, two backward
// launches, no [Tq, Tk] tensor in HBM.
//
// bf16 (MFMA): flash structure with the band folded in.  For one (batch, head) a wave owns 16 queries; scores are produced
// as S^T = K Q^T (rows = keys, cols = queries) so the lane that owns query n keeps its softmax statistics, and P^T is
// directly the B operand of O^T += V^T P^T (A operand through ds_read_b64_tr_b16).  The relative part (template REL):
//   * logits: qe[i][r] = q_i . Ek[r] is ONE extra 16x16 MFMA tile per query tile (rows = the 2w+1 <= 16 offsets), parked
//     in LDS and added to the scores of keys j = i + r - w;
//   * values: the (at most 2w+1) probabilities of a query that sit on the band are recorded as raw scores while the key
//     loop runs, turned into probabilities with the final max / sum, and applied as one more MFMA against Ev.
// fp32: the north_star's 1e-3 parity path -- plain VALU kernels (one wave per query / per key, scores of a row in LDS),
// same masks, same dropout hash, same outputs; nothing about them is tuned, they exist so that the fp32 golden runs go
// through this library and not through torch.
// Key padding: keys j >= lens_k[b] are excluded (the reference's -1e4 / -inf fill underflows to exactly 0 in fp32
// whenever the row has a live key); query rows i >= lens_q[b] are written as zeros (see enc_ops.hip for why that is
// equivalent).  Dropout: keep(b, h, i, j) = hash(*seed_dev, site, ...) as in enc_ops.hip -- regenerated in both backward
// kernels.
//
// Backward: mha_bwd_dq (same ownership as forward: dQ, and the 2 x [2w+1][D] embedding gradients) and mha_bwd_dkv (a
// wave owns 16 keys, S = Q K^T layout: dK, dV), both recompute P from the saved log-sum-exp.
#include "evt_common.h"
#include "../../include/evt.h"

namespace {

struct RP {
  const void* q; const void* k; const void* v;              // q [B][Tq][ldq], k / v [B][Tk][ldk] rows, head h at column h*D
  const void* o; const void* d_o;                           // [B][Tq][ldo]
  void* out; void* dq; void* dk; void* dv;
  const float* ek; const float* ev;                         // [Hr][R][D] fp32 parameters (REL)
  float* dek; float* dev;                                   // fp32 accumulators (+=)
  const int* lens_q; const int* lens_k;                     // [B] or null
  float* lse; float* delta;                                 // [B*H][Tq]
  int B, Tq, Tk, H, D, Hr, R, w;
  long ldq, ldk, ldo;
  float scale;
  unsigned thr; float keep_scale;                           // dropout: keep iff hash >= thr (0 = off)
  const unsigned* seed_dev; unsigned site;
};

__device__ __forceinline__ unsigned mix32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ unsigned drop_key(const RP& p) {
  return mix32((p.seed_dev ? *p.seed_dev : 0u) * 0x9E3779B1u + p.site * 0x85EBCA77u + 0x27D4EB2Fu);
}
// per (b*H+h, query) row key, then per key column
__device__ __forceinline__ unsigned drop_row(unsigned key, unsigned bh, int qi) {
  return mix32(key ^ (bh * 0x9E3779B1u) ^ ((unsigned)qi * 0x85EBCA77u));
}
__device__ __forceinline__ float drop_mult(const RP& p, unsigned row, int kj) {
  if (p.thr == 0u) return 1.f;
  return mix32(row + (unsigned)kj * 0xC2B2AE35u) >= p.thr ? p.keep_scale : 0.f;
}

__device__ __forceinline__ h16x8 tr2(const h16_t* p0, const h16_t* p1) {
  const unsigned a0 = (unsigned)(uintptr_t)p0, a1 = (unsigned)(uintptr_t)p1;
  uint2 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(lo), "=&v"(hi) : "v"(a0), "v"(a1) : "memory");
  union { uint4 u; h16x8 v; } r;
  r.u = make_uint4(lo.x, lo.y, hi.x, hi.y);
  return r.v;
}
__device__ __forceinline__ h16x8 pack8(const float* p) {
  union { h16x8 v; h16_t e[8]; } r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.e[i] = f2h(p[i]);
  return r.v;
}
__device__ __forceinline__ h16x8 ld8(const h16_t* p) { return *reinterpret_cast<const h16x8*>(p); }
__device__ __forceinline__ int slot32(int g, int e) { return e < 4 ? g * 4 + e : 16 + g * 4 + (e - 4); }

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

constexpr int RB = 17;   // pitch (floats) of the per-query band arrays [16 queries][16 offsets]

// stage `nrows` rows of a [.][ld] bf16 matrix (head slice of D columns) into an LDS tile; rows outside [0, limit) -> 0
template <int D, int PITCH>
__device__ __forceinline__ void stage_rows(h16_t* dst, const h16_t* src, long ld, int row0, int nrows, int limit) {
  constexpr int PPR = D / 8;
  for (int i = threadIdx.x; i < nrows * PPR; i += 256) {
    const int r = i / PPR, c8 = i - r * PPR;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row0 + r < limit) v = *reinterpret_cast<const uint4*>(src + (long)(row0 + r) * ld + c8 * 8);
    *reinterpret_cast<uint4*>(dst + r * PITCH + c8 * 8) = v;
  }
}
// the fp32 [R][D] embedding of this head -> bf16 LDS tile of `nrows` rows (rows >= R zero)
template <int D, int PITCH>
__device__ __forceinline__ void stage_emb(h16_t* dst, const float* src, int R, int nrows) {
  for (int i = threadIdx.x; i < nrows * D; i += 256) {
    const int r = i / D, c = i - r * D;
    dst[r * PITCH + c] = r < R ? f2h(src[r * D + c]) : (h16_t)0;
  }
}

// ---------------------------------------------------------------------------------------------------------
// forward: block = 64 queries of one (b, h), 4 waves x 16 queries; keys in tiles of 32
// ---------------------------------------------------------------------------------------------------------
template <int DK, bool REL>
__global__ __launch_bounds__(256) void mha_fwd_bf16(RP p) {
  constexpr int D = 32 * DK, PITCH = D + 8, NDT = D / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  h16_t* Qs = reinterpret_cast<h16_t*>(smem);            // [64][PITCH]
  h16_t* Ks = Qs + 64 * PITCH;                            // [32][PITCH]
  h16_t* Vs = Ks + 32 * PITCH;                            // [32][PITCH]
  h16_t* Eks = Vs + 32 * PITCH;                           // REL: [16][PITCH]
  h16_t* Evs = Eks + 16 * PITCH;                          // REL: [32][PITCH]
  float* qe_l = reinterpret_cast<float*>(Evs + 32 * PITCH);   // REL: [4][16][RB]
  float* sb_l = qe_l + 4 * 16 * RB;                        // REL: [4][16][RB] raw band scores
  h16_t* rw_l = reinterpret_cast<h16_t*>(sb_l + 4 * 16 * RB);   // REL: [4][16][40]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
  const int qb0 = blockIdx.x * 64;
  const int lenq = p.lens_q ? min(p.lens_q[b], p.Tq) : p.Tq;
  const int lenk = p.lens_k ? min(p.lens_k[b], p.Tk) : p.Tk;
  h16_t* O = (h16_t*)p.out + (long)b * p.Tq * p.ldo + h * D;
  float* LSE = p.lse + (long)bh * p.Tq;
  if (qb0 >= lenq) {   // block of padded queries: zeros
    for (int i = tid; i < 64 * (D / 8); i += 256) {
      const int r = i / (D / 8), c8 = i - r * (D / 8);
      if (qb0 + r < p.Tq) *reinterpret_cast<uint4*>(O + (long)(qb0 + r) * p.ldo + c8 * 8) = make_uint4(0, 0, 0, 0);
    }
    if (tid < 64 && qb0 + tid < p.Tq) LSE[qb0 + tid] = 0.f;
    return;
  }
  const h16_t* Q = (const h16_t*)p.q + (long)b * p.Tq * p.ldq + h * D;
  const h16_t* K = (const h16_t*)p.k + (long)b * p.Tk * p.ldk + h * D;
  const h16_t* V = (const h16_t*)p.v + (long)b * p.Tk * p.ldk + h * D;
  const int hr = REL ? h % p.Hr : 0;
  stage_rows<D, PITCH>(Qs, Q, p.ldq, qb0, 64, lenq);
  if constexpr (REL) {
    stage_emb<D, PITCH>(Eks, p.ek + (long)hr * p.R * D, p.R, 16);
    stage_emb<D, PITCH>(Evs, p.ev + (long)hr * p.R * D, p.R, 32);
  }
  __syncthreads();

  const int q0 = qb0 + wave * 16, qi = q0 + n;
  h16x8 qf[DK];
#pragma unroll
  for (int s = 0; s < DK; ++s) qf[s] = ld8(Qs + (wave * 16 + n) * PITCH + s * 32 + g * 8);
  float* qe = qe_l + wave * 16 * RB;
  float* sb = sb_l + wave * 16 * RB;
  if constexpr (REL) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < DK; ++s)
      acc = EVT_MFMA_16x16x32(ld8(Eks + n * PITCH + s * 32 + g * 8), qf[s], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) { qe[n * RB + g * 4 + r] = acc[r] * p.scale; sb[n * RB + g * 4 + r] = -INFINITY; }
  }
  const unsigned drow = drop_row(drop_key(p), (unsigned)bh, qi);
  float m_run = -INFINITY, l_run = 0.f;
  f32x4 ot[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) ot[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < lenk; k0 += 32) {
    __syncthreads();
    stage_rows<D, PITCH>(Ks, K, p.ldk, k0, 32, lenk);
    stage_rows<D, PITCH>(Vs, V, p.ldk, k0, 32, lenk);
    __syncthreads();
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < DK; ++s) {
      s0 = EVT_MFMA_16x16x32(ld8(Ks + n * PITCH + s * 32 + g * 8), qf[s], s0, 0, 0, 0);
      s1 = EVT_MFMA_16x16x32(ld8(Ks + (16 + n) * PITCH + s * 32 + g * 8), qf[s], s1, 0, 0, 0);
    }
    const bool band = REL && (k0 <= q0 + 15 + p.w) && (k0 + 31 >= q0 - p.w);   // wave-uniform
    float sc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int kj = k0 + slot32(g, e);
      float v = (e < 4 ? s0[e] : s1[e - 4]) * p.scale;
      if constexpr (REL) {
        if (band) {
          const int rr = kj - qi + p.w;
          if (rr >= 0 && rr < p.R) {
            v += qe[n * RB + rr];
            if (kj < lenk) sb[n * RB + rr] = v;
          }
        }
      }
      sc[e] = kj < lenk ? v : -INFINITY;
    }
    float mx = fmaxf(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])), fmaxf(fmaxf(sc[4], sc[5]), fmaxf(sc[6], sc[7])));
    mx = quad_max(mx);
    const float mn = fmaxf(m_run, mx);       // finite: tile k0 = 0 always holds key 0 < lenk
    const float alpha = __expf(m_run - mn);
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = __expf(sc[e] - mn); sum += sc[e]; }
    sum = quad_sum(sum);
    l_run = l_run * alpha + sum;
    m_run = mn;
    if (p.thr) {
#pragma unroll
      for (int e = 0; e < 8; ++e) sc[e] *= drop_mult(p, drow, k0 + slot32(g, e));
    }
    const h16x8 pf = pack8(sc);
    const h16_t* vrow = Vs + (g * 4 + (n >> 2)) * PITCH + 4 * (n & 3);
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) ot[dt][r] *= alpha;
      ot[dt] = EVT_MFMA_16x16x32(tr2(vrow + dt * 16, vrow + 16 * PITCH + dt * 16), pf, ot[dt], 0, 0, 0);
    }
  }
  const float inv = l_run > 0.f ? 1.f / l_run : 0.f;      // an item without a single live key: zeros
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int r = 0; r < 4; ++r) ot[dt][r] *= inv;
  if constexpr (REL) {
    // relative values: the band probabilities with the final statistics, one MFMA step against Ev
    __syncthreads();
    h16_t* rw = rw_l + wave * 16 * 40;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rr = g * 4 + r;
      const float s = sb[n * RB + rr];
      float pv = 0.f;
      if (rr < p.R && s > -INFINITY) pv = __expf(s - m_run) * inv * drop_mult(p, drow, qi + rr - p.w);
      rw[n * 40 + rr] = f2h(pv);
      rw[n * 40 + 16 + rr] = (h16_t)0;
    }
    __syncthreads();
    const h16x8 rf = ld8(rw + n * 40 + g * 8);
    const h16_t* erow = Evs + (g * 8 + (n >> 2)) * PITCH + 4 * (n & 3);
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
      ot[dt] = EVT_MFMA_16x16x32(tr2(erow + dt * 16, erow + 4 * PITCH + dt * 16), rf, ot[dt], 0, 0, 0);
  }
  if (qi < p.Tq) {
    const bool live = qi < lenq;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      h16_t o4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) o4[r] = live ? f2h(ot[dt][r]) : (h16_t)0;
      *reinterpret_cast<uint2*>(O + (long)qi * p.ldo + dt * 16 + g * 4) = *reinterpret_cast<uint2*>(o4);
    }
    if (g == 0) LSE[qi] = (live && l_run > 0.f) ? m_run + __logf(l_run) : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------------------
// dQ + embedding gradients: same ownership as forward
// ---------------------------------------------------------------------------------------------------------
template <int DK, bool REL>
__global__ __launch_bounds__(256) void mha_bwd_dq_bf16(RP p) {
  constexpr int D = 32 * DK, PITCH = D + 8, NDT = D / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  h16_t* Qs = reinterpret_cast<h16_t*>(smem);            // [64][PITCH]
  h16_t* dOs = Qs + 64 * PITCH;                           // [64][PITCH]
  h16_t* Ks = dOs + 64 * PITCH;                           // [32][PITCH]
  h16_t* Vs = Ks + 32 * PITCH;                            // [32][PITCH]
  float* dl_l = reinterpret_cast<float*>(Vs + 32 * PITCH); // [64] delta
  h16_t* Eks = reinterpret_cast<h16_t*>(dl_l + 64);      // REL: [32][PITCH] (rows >= R zero)
  h16_t* Evs = Eks + 32 * PITCH;                          // REL: [16][PITCH]
  float* qe_l = reinterpret_cast<float*>(Evs + 16 * PITCH);   // REL: [64][RB]
  float* de_l = qe_l + 64 * RB;                            // REL: [64][RB]  dO . Ev[r]
  float* ds_l = de_l + 64 * RB;                            // REL: [64][RB]  dS on the band
  float* pb_l = ds_l + 64 * RB;                            // REL: [64][RB]  dropped P on the band
  h16_t* dw_l = reinterpret_cast<h16_t*>(pb_l + 64 * RB);   // REL: [4][16][40]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
  const int qb0 = blockIdx.x * 64;
  const int lenq = p.lens_q ? min(p.lens_q[b], p.Tq) : p.Tq;
  const int lenk = p.lens_k ? min(p.lens_k[b], p.Tk) : p.Tk;
  h16_t* dQ = (h16_t*)p.dq + (long)b * p.Tq * p.ldq + h * D;
  float* DL = p.delta + (long)bh * p.Tq;
  if (qb0 >= lenq) {
    for (int i = tid; i < 64 * (D / 8); i += 256) {
      const int r = i / (D / 8), c8 = i - r * (D / 8);
      if (qb0 + r < p.Tq) *reinterpret_cast<uint4*>(dQ + (long)(qb0 + r) * p.ldq + c8 * 8) = make_uint4(0, 0, 0, 0);
    }
    if (tid < 64 && qb0 + tid < p.Tq) DL[qb0 + tid] = 0.f;
    return;
  }
  const h16_t* Q = (const h16_t*)p.q + (long)b * p.Tq * p.ldq + h * D;
  const h16_t* K = (const h16_t*)p.k + (long)b * p.Tk * p.ldk + h * D;
  const h16_t* V = (const h16_t*)p.v + (long)b * p.Tk * p.ldk + h * D;
  const h16_t* Og = (const h16_t*)p.o + (long)b * p.Tq * p.ldo + h * D;
  const h16_t* dOg = (const h16_t*)p.d_o + (long)b * p.Tq * p.ldo + h * D;
  const int hr = REL ? h % p.Hr : 0;
  stage_rows<D, PITCH>(Qs, Q, p.ldq, qb0, 64, lenq);
  stage_rows<D, PITCH>(dOs, dOg, p.ldo, qb0, 64, lenq);
  if constexpr (REL) {
    stage_emb<D, PITCH>(Eks, p.ek + (long)hr * p.R * D, p.R, 32);
    stage_emb<D, PITCH>(Evs, p.ev + (long)hr * p.R * D, p.R, 16);
    for (int i = tid; i < 64 * RB; i += 256) { ds_l[i] = 0.f; pb_l[i] = 0.f; }
  }
  if (tid < 64) dl_l[tid] = 0.f;
  __syncthreads();
  // delta_i = dO_i . O_i : 4 lanes per row, 16-byte pieces
  {
    const int r = tid >> 2, part = tid & 3;
    float acc = 0.f;
    if (qb0 + r < lenq) {
      for (int c8 = part; c8 < D / 8; c8 += 4) {
        const uint4 ov = *reinterpret_cast<const uint4*>(Og + (long)(qb0 + r) * p.ldo + c8 * 8);
        const h16_t* po = reinterpret_cast<const h16_t*>(&ov);
        const h16_t* pd = dOs + r * PITCH + c8 * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += h2f(po[e]) * h2f(pd[e]);
      }
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    if (part == 0) { dl_l[r] = acc; if (qb0 + r < p.Tq) DL[qb0 + r] = acc; }
  }
  const int q0 = qb0 + wave * 16, qi = q0 + n;
  const int ql = wave * 16 + n;       // row inside the block tiles
  h16x8 qf[DK], dof[DK];
#pragma unroll
  for (int s = 0; s < DK; ++s) {
    qf[s] = ld8(Qs + ql * PITCH + s * 32 + g * 8);
    dof[s] = ld8(dOs + ql * PITCH + s * 32 + g * 8);
  }
  if constexpr (REL) {
    f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < DK; ++s) {
      a1 = EVT_MFMA_16x16x32(ld8(Eks + n * PITCH + s * 32 + g * 8), qf[s], a1, 0, 0, 0);
      a2 = EVT_MFMA_16x16x32(ld8(Evs + n * PITCH + s * 32 + g * 8), dof[s], a2, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { qe_l[ql * RB + g * 4 + r] = a1[r] * p.scale; de_l[ql * RB + g * 4 + r] = a2[r]; }
  }
  __syncthreads();
  const float lse = qi < lenq ? p.lse[(long)bh * p.Tq + qi] : 0.f;
  const float dlt = dl_l[ql];
  const unsigned drow = drop_row(drop_key(p), (unsigned)bh, qi);
  f32x4 dqt[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) dqt[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < lenk; k0 += 32) {
    __syncthreads();
    stage_rows<D, PITCH>(Ks, K, p.ldk, k0, 32, lenk);
    stage_rows<D, PITCH>(Vs, V, p.ldk, k0, 32, lenk);
    __syncthreads();
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f}, d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < DK; ++s) {
      s0 = EVT_MFMA_16x16x32(ld8(Ks + n * PITCH + s * 32 + g * 8), qf[s], s0, 0, 0, 0);
      s1 = EVT_MFMA_16x16x32(ld8(Ks + (16 + n) * PITCH + s * 32 + g * 8), qf[s], s1, 0, 0, 0);
      d0 = EVT_MFMA_16x16x32(ld8(Vs + n * PITCH + s * 32 + g * 8), dof[s], d0, 0, 0, 0);
      d1 = EVT_MFMA_16x16x32(ld8(Vs + (16 + n) * PITCH + s * 32 + g * 8), dof[s], d1, 0, 0, 0);
    }
    const bool band = REL && (k0 <= q0 + 15 + p.w) && (k0 + 31 >= q0 - p.w);
    float ds[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int kj = k0 + slot32(g, e);
      float v = (e < 4 ? s0[e] : s1[e - 4]) * p.scale;
      float dp = (e < 4 ? d0[e] : d1[e - 4]);
      const int rr = kj - qi + p.w;
      const bool onb = band && rr >= 0 && rr < p.R;
      if constexpr (REL) {
        if (onb) { v += qe_l[ql * RB + rr]; dp += de_l[ql * RB + rr]; }
      }
      const bool ok = kj < lenk && qi < lenq;
      const float pr = ok ? __expf(v - lse) : 0.f;
      const float mult = drop_mult(p, drow, kj);
      const float dsv = pr * (dp * mult - dlt);
      if constexpr (REL) {
        if (onb && ok) { ds_l[ql * RB + rr] = dsv; pb_l[ql * RB + rr] = pr * mult; }
      }
      ds[e] = dsv;
    }
    const h16x8 dsf = pack8(ds);
    const h16_t* krow = Ks + (g * 4 + (n >> 2)) * PITCH + 4 * (n & 3);
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
      dqt[dt] = EVT_MFMA_16x16x32(tr2(krow + dt * 16, krow + 16 * PITCH + dt * 16), dsf, dqt[dt], 0, 0, 0);
  }
  if constexpr (REL) {
    // band part of dQ: dS[i, i+r-w] * Ek[r]
    __syncthreads();
    h16_t* dw = dw_l + wave * 16 * 40;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      dw[n * 40 + g * 4 + r] = f2h(ds_l[ql * RB + g * 4 + r]);
      dw[n * 40 + 16 + g * 4 + r] = (h16_t)0;
    }
    __syncthreads();
    const h16x8 rf = ld8(dw + n * 40 + g * 8);
    const h16_t* erow = Eks + (g * 8 + (n >> 2)) * PITCH + 4 * (n & 3);
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
      dqt[dt] = EVT_MFMA_16x16x32(tr2(erow + dt * 16, erow + 4 * PITCH + dt * 16), rf, dqt[dt], 0, 0, 0);
  }
  if (qi < p.Tq) {
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      h16_t o4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) o4[r] = qi < lenq ? f2h(dqt[dt][r] * p.scale) : (h16_t)0;
      *reinterpret_cast<uint2*>(dQ + (long)qi * p.ldq + dt * 16 + g * 4) = *reinterpret_cast<uint2*>(o4);
    }
  }
  if constexpr (REL) {
    // embedding gradients of this block: dEk[r][d] += scale * sum_q dS_band[q][r] Q[q][d];  dEv[r][d] += sum_q P_band[q][r] dO[q][d]
    float* dek = p.dek + (long)hr * p.R * D;
    float* dev = p.dev + (long)hr * p.R * D;
    for (int i = tid; i < p.R * D; i += 256) {
      const int r = i / D, c = i - r * D;
      float a1 = 0.f, a2 = 0.f;
      for (int q = 0; q < 64; ++q) {
        a1 += ds_l[q * RB + r] * h2f(Qs[q * PITCH + c]);
        a2 += pb_l[q * RB + r] * h2f(dOs[q * PITCH + c]);
      }
      atomicAdd(dek + i, a1 * p.scale);
      atomicAdd(dev + i, a2);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// dK, dV: block = 64 keys of one (b, h), a wave owns 16 keys; S = Q K^T layout (rows = queries, cols = keys);
// queries in tiles of 32.  P / dS are directly the B operands of dV^T += dO^T P and dK^T += Q^T dS.
// ---------------------------------------------------------------------------------------------------------
template <int DK, bool REL>
__global__ __launch_bounds__(256) void mha_bwd_dkv_bf16(RP p) {
  constexpr int D = 32 * DK, PITCH = D + 8, NDT = D / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  h16_t* Ks = reinterpret_cast<h16_t*>(smem);            // [64][PITCH]
  h16_t* Vs = Ks + 64 * PITCH;                            // [64][PITCH]
  h16_t* Qs = Vs + 64 * PITCH;                            // [32][PITCH]
  h16_t* dOs = Qs + 32 * PITCH;                           // [32][PITCH]
  float* ls_l = reinterpret_cast<float*>(dOs + 32 * PITCH);   // [32] lse
  float* dl_l = ls_l + 32;                                 // [32] delta
  h16_t* Eks = reinterpret_cast<h16_t*>(dl_l + 32);      // REL: [16][PITCH]
  h16_t* Evs = Eks + 16 * PITCH;                          // REL: [16][PITCH]
  float* qe_l = reinterpret_cast<float*>(Evs + 16 * PITCH);   // REL: [32][RB]
  float* de_l = qe_l + 32 * RB;                            // REL: [32][RB]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
  const int kb0 = blockIdx.x * 64;
  const int lenq = p.lens_q ? min(p.lens_q[b], p.Tq) : p.Tq;
  const int lenk = p.lens_k ? min(p.lens_k[b], p.Tk) : p.Tk;
  h16_t* dKg = (h16_t*)p.dk + (long)b * p.Tk * p.ldk + h * D;
  h16_t* dVg = (h16_t*)p.dv + (long)b * p.Tk * p.ldk + h * D;
  if (kb0 >= lenk) {
    for (int i = tid; i < 64 * (D / 8); i += 256) {
      const int r = i / (D / 8), c8 = i - r * (D / 8);
      if (kb0 + r < p.Tk) {
        *reinterpret_cast<uint4*>(dKg + (long)(kb0 + r) * p.ldk + c8 * 8) = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(dVg + (long)(kb0 + r) * p.ldk + c8 * 8) = make_uint4(0, 0, 0, 0);
      }
    }
    return;
  }
  const h16_t* Q = (const h16_t*)p.q + (long)b * p.Tq * p.ldq + h * D;
  const h16_t* K = (const h16_t*)p.k + (long)b * p.Tk * p.ldk + h * D;
  const h16_t* V = (const h16_t*)p.v + (long)b * p.Tk * p.ldk + h * D;
  const h16_t* dOg = (const h16_t*)p.d_o + (long)b * p.Tq * p.ldo + h * D;
  const int hr = REL ? h % p.Hr : 0;
  stage_rows<D, PITCH>(Ks, K, p.ldk, kb0, 64, lenk);
  stage_rows<D, PITCH>(Vs, V, p.ldk, kb0, 64, lenk);
  if constexpr (REL) {
    stage_emb<D, PITCH>(Eks, p.ek + (long)hr * p.R * D, p.R, 16);
    stage_emb<D, PITCH>(Evs, p.ev + (long)hr * p.R * D, p.R, 16);
  }
  __syncthreads();
  const int k0w = kb0 + wave * 16, kj = k0w + n;     // my key (MFMA column)
  h16x8 kf[DK], vf[DK];
#pragma unroll
  for (int s = 0; s < DK; ++s) {
    kf[s] = ld8(Ks + (wave * 16 + n) * PITCH + s * 32 + g * 8);
    vf[s] = ld8(Vs + (wave * 16 + n) * PITCH + s * 32 + g * 8);
  }
  const unsigned dkey = drop_key(p);
  f32x4 dkt[NDT], dvt[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) { dkt[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dvt[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  for (int q0 = 0; q0 < lenq; q0 += 32) {
    __syncthreads();
    stage_rows<D, PITCH>(Qs, Q, p.ldq, q0, 32, lenq);
    stage_rows<D, PITCH>(dOs, dOg, p.ldo, q0, 32, lenq);
    if (tid < 32) {
      const bool ok = q0 + tid < lenq;
      ls_l[tid] = ok ? p.lse[(long)bh * p.Tq + q0 + tid] : 0.f;
      dl_l[tid] = ok ? p.delta[(long)bh * p.Tq + q0 + tid] : 0.f;
    }
    __syncthreads();
    // band tables of this query tile (block-uniform test): waves 0/1 -> q.Ek, waves 2/3 -> dO.Ev, 16 queries each
    const bool band = REL && (kb0 <= q0 + 31 + p.w) && (kb0 + 63 >= q0 - p.w);
    if constexpr (REL) {
      if (band) {
        const int tl = wave & 1;
        const h16_t* As = (wave < 2 ? Qs : dOs) + (tl * 16 + n) * PITCH + g * 8;      // A: m = query
        const h16_t* Bs = (wave < 2 ? Eks : Evs) + n * PITCH + g * 8;                  // B: n = offset r
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < DK; ++s) acc = EVT_MFMA_16x16x32(ld8(As + s * 32), ld8(Bs + s * 32), acc, 0, 0, 0);
        float* dst = wave < 2 ? qe_l : de_l;          // result: lane (n = r, g) holds query tl*16 + g*4 + rr
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(tl * 16 + g * 4 + r) * RB + n] = wave < 2 ? acc[r] * p.scale : acc[r];
      }
      __syncthreads();
    }
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f}, d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < DK; ++s) {
      s0 = EVT_MFMA_16x16x32(ld8(Qs + n * PITCH + s * 32 + g * 8), kf[s], s0, 0, 0, 0);
      s1 = EVT_MFMA_16x16x32(ld8(Qs + (16 + n) * PITCH + s * 32 + g * 8), kf[s], s1, 0, 0, 0);
      d0 = EVT_MFMA_16x16x32(ld8(dOs + n * PITCH + s * 32 + g * 8), vf[s], d0, 0, 0, 0);
      d1 = EVT_MFMA_16x16x32(ld8(dOs + (16 + n) * PITCH + s * 32 + g * 8), vf[s], d1, 0, 0, 0);
    }
    float pd[8], ds[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int qloc = slot32(g, e), qi = q0 + qloc;
      float v = (e < 4 ? s0[e] : s1[e - 4]) * p.scale;
      float dp = (e < 4 ? d0[e] : d1[e - 4]);
      if constexpr (REL) {
        const int rr = kj - qi + p.w;
        if (band && rr >= 0 && rr < p.R) { v += qe_l[qloc * RB + rr]; dp += de_l[qloc * RB + rr]; }
      }
      const bool ok = kj < lenk && qi < lenq;
      const float pr = ok ? __expf(v - ls_l[qloc]) : 0.f;
      const float mult = drop_mult(p, drop_row(dkey, (unsigned)bh, qi), kj);
      pd[e] = pr * mult;
      ds[e] = pr * (dp * mult - dl_l[qloc]);
    }
    const h16x8 pf = pack8(pd), dsf = pack8(ds);
    const h16_t* orow = dOs + (g * 4 + (n >> 2)) * PITCH + 4 * (n & 3);
    const h16_t* qrow = Qs + (g * 4 + (n >> 2)) * PITCH + 4 * (n & 3);
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      dvt[dt] = EVT_MFMA_16x16x32(tr2(orow + dt * 16, orow + 16 * PITCH + dt * 16), pf, dvt[dt], 0, 0, 0);
      dkt[dt] = EVT_MFMA_16x16x32(tr2(qrow + dt * 16, qrow + 16 * PITCH + dt * 16), dsf, dkt[dt], 0, 0, 0);
    }
  }
  if (kj < p.Tk) {
    const bool live = kj < lenk;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      h16_t k4[4], v4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        k4[r] = live ? f2h(dkt[dt][r] * p.scale) : (h16_t)0;
        v4[r] = live ? f2h(dvt[dt][r]) : (h16_t)0;
      }
      *reinterpret_cast<uint2*>(dKg + (long)kj * p.ldk + dt * 16 + g * 4) = *reinterpret_cast<uint2*>(k4);
      *reinterpret_cast<uint2*>(dVg + (long)kj * p.ldk + dt * 16 + g * 4) = *reinterpret_cast<uint2*>(v4);
    }
  }
}

template <int DK, bool REL> constexpr size_t fwd_lds() {
  constexpr int PITCH = 32 * DK + 8;
  return (size_t)(64 + 32 + 32) * PITCH * 2 + (REL ? (size_t)(16 + 32) * PITCH * 2 + 2 * 4 * 16 * RB * 4 + 4 * 16 * 40 * 2 : 0);
}
template <int DK, bool REL> constexpr size_t dq_lds() {
  constexpr int PITCH = 32 * DK + 8;
  return (size_t)(64 + 64 + 32 + 32) * PITCH * 2 + 64 * 4 +
         (REL ? (size_t)(32 + 16) * PITCH * 2 + 4 * 64 * RB * 4 + 4 * 16 * 40 * 2 : 0);
}
template <int DK, bool REL> constexpr size_t dkv_lds() {
  constexpr int PITCH = 32 * DK + 8;
  return (size_t)(64 + 64 + 32 + 32) * PITCH * 2 + 64 * 4 + (REL ? (size_t)(16 + 16) * PITCH * 2 + 2 * 32 * RB * 4 : 0);
}

// ---------------------------------------------------------------------------------------------------------
// fp32 (the 1e-3 parity path): one wave per query (forward, dQ) / per key (dK, dV), four per block; the scores of the
// wave's row live in LDS.  D % 4 == 0, D <= 128; rows 16-byte aligned.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dot_row(const float* a_lds, const float* g, int D) {
  float acc = 0.f;
  for (int d = 0; d < D; d += 4) {
    const float4 v = *reinterpret_cast<const float4*>(g + d);
    acc += a_lds[d] * v.x + a_lds[d + 1] * v.y + a_lds[d + 2] * v.z + a_lds[d + 3] * v.w;
  }
  return acc;
}

template <bool REL>
__global__ __launch_bounds__(256) void mha_fwd_f32(RP p, int tkp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int D = p.D;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
  const int lenq = p.lens_q ? min(p.lens_q[b], p.Tq) : p.Tq;
  const int lenk = p.lens_k ? min(p.lens_k[b], p.Tk) : p.Tk;
  const int qraw = blockIdx.x * 4 + wave;
  const bool inb = qraw < p.Tq;
  const int qi = inb ? qraw : p.Tq - 1;
  const bool live = inb && qi < lenq;
  float* qs = reinterpret_cast<float*>(smem) + wave * (D + tkp);   // q * scale
  float* sc = qs + D;                                               // [tkp]
  const float* Q = (const float*)p.q + ((long)b * p.Tq + qi) * p.ldq + h * D;
  const float* K = (const float*)p.k + (long)b * p.Tk * p.ldk + h * D;
  const float* V = (const float*)p.v + (long)b * p.Tk * p.ldk + h * D;
  const float* ek = REL ? p.ek + (long)(h % p.Hr) * p.R * D : nullptr;
  const float* ev = REL ? p.ev + (long)(h % p.Hr) * p.R * D : nullptr;
  for (int d = lane; d < D; d += 64) qs[d] = Q[d] * p.scale;
  __syncthreads();
  float m = -INFINITY;
  for (int j = lane; j < lenk; j += 64) {
    float acc = dot_row(qs, K + (long)j * p.ldk, D);
    if constexpr (REL) {
      const int rr = j - qi + p.w;
      if (rr >= 0 && rr < p.R) acc += dot_row(qs, ek + rr * D, D);
    }
    sc[j] = acc;
    m = fmaxf(m, acc);
  }
  m = wave_reduce_max(m);
  float l = 0.f;
  for (int j = lane; j < lenk; j += 64) { const float e = expf(sc[j] - m); sc[j] = e; l += e; }
  l = wave_reduce_sum(l);
  const float inv = l > 0.f ? 1.f / l : 0.f;
  const unsigned drow = drop_row(drop_key(p), (unsigned)bh, qi);
  for (int j = lane; j < lenk; j += 64) sc[j] = sc[j] * inv * drop_mult(p, drow, j);
  __syncthreads();
  float* O = (float*)p.out + ((long)b * p.Tq + qi) * p.ldo + h * D;
  for (int d = lane; d < D; d += 64) {
    float acc = 0.f;
    for (int j = 0; j < lenk; ++j) acc += sc[j] * V[(long)j * p.ldk + d];
    if constexpr (REL) {
      for (int r = 0; r < p.R; ++r) {
        const int j = qi + r - p.w;
        if (j >= 0 && j < lenk) acc += sc[j] * ev[r * D + d];
      }
    }
    if (inb) O[d] = live ? acc : 0.f;
  }
  if (lane == 0 && inb) p.lse[(long)bh * p.Tq + qi] = (live && l > 0.f) ? m + logf(l) : 0.f;
}

template <bool REL>
__global__ __launch_bounds__(256) void mha_bwd_dq_f32(RP p, int tkp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int D = p.D;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
  const int lenq = p.lens_q ? min(p.lens_q[b], p.Tq) : p.Tq;
  const int lenk = p.lens_k ? min(p.lens_k[b], p.Tk) : p.Tk;
  const int qraw = blockIdx.x * 4 + wave;
  const bool inb = qraw < p.Tq;
  const int qi = inb ? qraw : p.Tq - 1;
  const bool live = inb && qi < lenq;
  float* qr = reinterpret_cast<float*>(smem) + wave * (2 * D + tkp + 32);   // raw q
  float* dof = qr + D;                                                        // dO row
  float* sc = dof + D;                                                        // [tkp] dS
  float* pbn = sc + tkp;                                                      // [16] dropped P on the band (REL)
  const float* Q = (const float*)p.q + ((long)b * p.Tq + qi) * p.ldq + h * D;
  const float* K = (const float*)p.k + (long)b * p.Tk * p.ldk + h * D;
  const float* V = (const float*)p.v + (long)b * p.Tk * p.ldk + h * D;
  const float* Og = (const float*)p.o + ((long)b * p.Tq + qi) * p.ldo + h * D;
  const float* dOg = (const float*)p.d_o + ((long)b * p.Tq + qi) * p.ldo + h * D;
  const float* ek = REL ? p.ek + (long)(h % p.Hr) * p.R * D : nullptr;
  const float* ev = REL ? p.ev + (long)(h % p.Hr) * p.R * D : nullptr;
  float dl = 0.f;
  for (int d = lane; d < D; d += 64) {
    const float q = Q[d], g = live ? dOg[d] : 0.f;
    qr[d] = q; dof[d] = g;
    dl += g * (live ? Og[d] : 0.f);
  }
  if (lane < 16) pbn[lane] = 0.f;
  dl = wave_reduce_sum(dl);
  __syncthreads();
  const float lse = live ? p.lse[(long)bh * p.Tq + qi] : 0.f;
  const unsigned drow = drop_row(drop_key(p), (unsigned)bh, qi);
  for (int j = lane; j < lenk; j += 64) {
    float s = dot_row(qr, K + (long)j * p.ldk, D);
    float dp = dot_row(dof, V + (long)j * p.ldk, D);
    int rr = -1;
    if constexpr (REL) {
      rr = j - qi + p.w;
      if (rr >= 0 && rr < p.R) { s += dot_row(qr, ek + rr * D, D); dp += dot_row(dof, ev + rr * D, D); } else rr = -1;
    }
    const float pr = live ? expf(s * p.scale - lse) : 0.f;
    const float mult = drop_mult(p, drow, j);
    sc[j] = pr * (dp * mult - dl);
    if (rr >= 0) pbn[rr] = pr * mult;
  }
  __syncthreads();
  float* dQ = (float*)p.dq + ((long)b * p.Tq + qi) * p.ldq + h * D;
  for (int d = lane; d < D; d += 64) {
    float acc = 0.f;
    for (int j = 0; j < lenk; ++j) acc += sc[j] * K[(long)j * p.ldk + d];
    if constexpr (REL) {
      float* dek = p.dek + (long)(h % p.Hr) * p.R * D;
      float* dev = p.dev + (long)(h % p.Hr) * p.R * D;
      for (int r = 0; r < p.R; ++r) {
        const int j = qi + r - p.w;
        if (j >= 0 && j < lenk) {
          acc += sc[j] * ek[r * D + d];
          if (live) {
            atomicAdd(dek + r * D + d, sc[j] * qr[d] * p.scale);
            atomicAdd(dev + r * D + d, pbn[r] * dof[d]);
          }
        }
      }
    }
    if (inb) dQ[d] = live ? acc * p.scale : 0.f;
  }
  if (lane == 0 && inb) p.delta[(long)bh * p.Tq + qi] = dl;
}

template <bool REL>
__global__ __launch_bounds__(256) void mha_bwd_dkv_f32(RP p, int tqp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int D = p.D;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
  const int lenq = p.lens_q ? min(p.lens_q[b], p.Tq) : p.Tq;
  const int lenk = p.lens_k ? min(p.lens_k[b], p.Tk) : p.Tk;
  const int kraw = blockIdx.x * 4 + wave;
  const bool inb = kraw < p.Tk;
  const int kj = inb ? kraw : p.Tk - 1;
  const bool live = inb && kj < lenk;
  float* kr = reinterpret_cast<float*>(smem) + wave * (2 * D + 2 * tqp);
  float* vr = kr + D;
  float* pd = vr + D;       // [tqp] dropped P
  float* ds = pd + tqp;     // [tqp] dS
  const float* Q = (const float*)p.q + (long)b * p.Tq * p.ldq + h * D;
  const float* dOg = (const float*)p.d_o + (long)b * p.Tq * p.ldo + h * D;
  const float* Kr = (const float*)p.k + ((long)b * p.Tk + kj) * p.ldk + h * D;
  const float* Vr = (const float*)p.v + ((long)b * p.Tk + kj) * p.ldk + h * D;
  const float* ek = REL ? p.ek + (long)(h % p.Hr) * p.R * D : nullptr;
  const float* ev = REL ? p.ev + (long)(h % p.Hr) * p.R * D : nullptr;
  for (int d = lane; d < D; d += 64) { kr[d] = Kr[d]; vr[d] = Vr[d]; }
  __syncthreads();
  const unsigned dkey = drop_key(p);
  for (int i = lane; i < lenq; i += 64) {
    const float* qrow = Q + (long)i * p.ldq;
    const float* orow = dOg + (long)i * p.ldo;
    float s = dot_row(kr, qrow, D);
    float dp = dot_row(vr, orow, D);
    if constexpr (REL) {
      const int rr = kj - i + p.w;
      if (rr >= 0 && rr < p.R) {
        const float* er = ek + rr * D; const float* fr = ev + rr * D;
        float a1 = 0.f, a2 = 0.f;
        for (int d = 0; d < D; ++d) { a1 += qrow[d] * er[d]; a2 += orow[d] * fr[d]; }
        s += a1; dp += a2;
      }
    }
    const float pr = live ? expf(s * p.scale - p.lse[(long)bh * p.Tq + i]) : 0.f;
    const float mult = drop_mult(p, drop_row(dkey, (unsigned)bh, i), kj);
    pd[i] = pr * mult;
    ds[i] = pr * (dp * mult - p.delta[(long)bh * p.Tq + i]);
  }
  __syncthreads();
  float* dK = (float*)p.dk + ((long)b * p.Tk + kj) * p.ldk + h * D;
  float* dV = (float*)p.dv + ((long)b * p.Tk + kj) * p.ldk + h * D;
  for (int d = lane; d < D; d += 64) {
    float ak = 0.f, av = 0.f;
    for (int i = 0; i < lenq; ++i) { ak += ds[i] * Q[(long)i * p.ldq + d]; av += pd[i] * dOg[(long)i * p.ldo + d]; }
    if (inb) { dK[d] = live ? ak * p.scale : 0.f; dV[d] = live ? av : 0.f; }
  }
}

int check(const evt_mha_params* a) {
  if (!a || a->B <= 0 || a->Tq <= 0 || a->Tk <= 0 || a->H <= 0 || a->D <= 0) return EVT_EINVAL;
  if (a->dtype != EVT_DT_HALF && a->dtype != EVT_DT_F32) return EVT_EINVAL;
  if (a->dtype == EVT_DT_HALF ? (a->D % 32 != 0) : (a->D % 4 != 0)) return EVT_ENOTSUP;
  if (a->D > 128) return EVT_ENOTSUP;
  if (a->window >= 0) {
    if (2 * a->window + 1 > 16) return EVT_ENOTSUP;
    if (a->Tq != a->Tk || a->n_heads_rel <= 0) return EVT_EINVAL;     // relative positions: self-attention only
  }
  const int64_t hd = (int64_t)a->H * a->D;
  if (a->ldq < hd || a->ldk < hd || a->ldo < hd) return EVT_EINVAL;
  const int al = a->dtype == EVT_DT_HALF ? 8 : 4;                      // 16-byte row pieces
  if (a->ldq % al || a->ldk % al || a->ldo % al) return EVT_EINVAL;
  if (a->dropout_p < 0.f || a->dropout_p >= 1.f) return EVT_EINVAL;
  if (!(a->scale > 0.f)) return EVT_EINVAL;
  return EVT_OK;
}

RP make_rp(const evt_mha_params* a) {
  RP p{};
  p.B = a->B; p.Tq = a->Tq; p.Tk = a->Tk; p.H = a->H; p.D = a->D;
  const bool rel = a->window >= 0;
  p.Hr = rel ? a->n_heads_rel : 1; p.w = rel ? a->window : 0; p.R = rel ? 2 * a->window + 1 : 0;
  p.ldq = a->ldq; p.ldk = a->ldk; p.ldo = a->ldo;
  p.scale = a->scale;
  p.thr = a->dropout_p > 0.f ? (unsigned)fminf(a->dropout_p * 4294967296.f, 4294967040.f) : 0u;
  p.keep_scale = a->dropout_p > 0.f ? 1.f / (1.f - a->dropout_p) : 1.f;
  p.seed_dev = a->seed_dev; p.site = a->site;
  return p;
}

// dynamic LDS above the 64 KB default needs the attribute; raised monotonically per kernel
template <typename F>
int ensure_lds(F fn, size_t lds, size_t* have) {
  if (lds <= *have) return EVT_OK;
  if (lds > 160 * 1024) return EVT_ENOTSUP;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return EVT_ELAUNCH;
  *have = lds;
  return EVT_OK;
}

template <int DK, bool REL>
int launch_fwd_bf16(const RP& p, hipStream_t st) {
  static size_t have = 64 * 1024;
  constexpr size_t lds = fwd_lds<DK, REL>();
  if (int rc = ensure_lds(&mha_fwd_bf16<DK, REL>, lds, &have)) return rc;
  const dim3 grid((p.Tq + 63) / 64, p.B * p.H);
  hipLaunchKernelGGL((mha_fwd_bf16<DK, REL>), grid, dim3(256), lds, st, p);
  return EVT_OK;
}
template <int DK, bool REL>
int launch_bwd_bf16(const RP& p, hipStream_t st) {
  static size_t have_q = 64 * 1024, have_k = 64 * 1024;
  constexpr size_t lq = dq_lds<DK, REL>(), lk = dkv_lds<DK, REL>();
  if (int rc = ensure_lds(&mha_bwd_dq_bf16<DK, REL>, lq, &have_q)) return rc;
  if (int rc = ensure_lds(&mha_bwd_dkv_bf16<DK, REL>, lk, &have_k)) return rc;
  hipLaunchKernelGGL((mha_bwd_dq_bf16<DK, REL>), dim3((p.Tq + 63) / 64, p.B * p.H), dim3(256), lq, st, p);
  hipLaunchKernelGGL((mha_bwd_dkv_bf16<DK, REL>), dim3((p.Tk + 63) / 64, p.B * p.H), dim3(256), lk, st, p);
  return EVT_OK;
}
template <bool REL>
int launch_fwd_f32(const RP& p, hipStream_t st) {
  static size_t have = 64 * 1024;
  const int tkp = (p.Tk + 3) / 4 * 4;
  const size_t lds = (size_t)4 * (p.D + tkp) * 4;
  if (int rc = ensure_lds(&mha_fwd_f32<REL>, lds, &have)) return rc;
  hipLaunchKernelGGL((mha_fwd_f32<REL>), dim3((p.Tq + 3) / 4, p.B * p.H), dim3(256), lds, st, p, tkp);
  return EVT_OK;
}
template <bool REL>
int launch_bwd_f32(const RP& p, hipStream_t st) {
  static size_t have_q = 64 * 1024, have_k = 64 * 1024;
  const int tkp = (p.Tk + 3) / 4 * 4, tqp = (p.Tq + 3) / 4 * 4;
  const size_t lq = (size_t)4 * (2 * p.D + tkp + 32) * 4, lk = (size_t)4 * (2 * p.D + 2 * tqp) * 4;
  if (int rc = ensure_lds(&mha_bwd_dq_f32<REL>, lq, &have_q)) return rc;
  if (int rc = ensure_lds(&mha_bwd_dkv_f32<REL>, lk, &have_k)) return rc;
  hipLaunchKernelGGL((mha_bwd_dq_f32<REL>), dim3((p.Tq + 3) / 4, p.B * p.H), dim3(256), lq, st, p, tkp);
  hipLaunchKernelGGL((mha_bwd_dkv_f32<REL>), dim3((p.Tk + 3) / 4, p.B * p.H), dim3(256), lk, st, p, tqp);
  return EVT_OK;
}

}  // namespace

extern "C" {

int evt_mha_fwd(const evt_mha_params* a, const void* q, const void* k, const void* v, const float* emb_k,
                const float* emb_v, const int32_t* lens_q, const int32_t* lens_k, void* out, float* lse, void* stream) {
  int rc = check(a);
  if (rc) return rc;
  const bool rel = a->window >= 0;
  if (!q || !k || !v || !out || !lse || (rel && (!emb_k || !emb_v))) return EVT_EINVAL;
  RP p = make_rp(a);
  p.q = q; p.k = k; p.v = v; p.ek = emb_k; p.ev = emb_v; p.lens_q = lens_q; p.lens_k = lens_k;
  p.out = out; p.lse = lse;
  hipStream_t st = (hipStream_t)stream;
  if (a->dtype == EVT_DT_F32) {
    evt_set_last_tag("mha_fwd_f32<%d>", (int)rel);
    rc = rel ? launch_fwd_f32<true>(p, st) : launch_fwd_f32<false>(p, st);
  } else {
    evt_set_last_tag("mha_fwd_bf16<%d, %d>", a->D / 32, (int)rel);
#define MHA_F(DK) rc = rel ? launch_fwd_bf16<DK, true>(p, st) : launch_fwd_bf16<DK, false>(p, st)
    switch (a->D / 32) { case 1: MHA_F(1); break; case 2: MHA_F(2); break; case 3: MHA_F(3); break; default: MHA_F(4); break; }
#undef MHA_F
  }
  return rc ? rc : evt_check_launch();
}

int evt_mha_bwd(const evt_mha_params* a, const void* q, const void* k, const void* v, const void* o, const void* d_o,
                const float* lse, const float* emb_k, const float* emb_v, const int32_t* lens_q, const int32_t* lens_k,
                void* dq, void* dk, void* dv, float* demb_k, float* demb_v, float* delta_ws, void* stream) {
  int rc = check(a);
  if (rc) return rc;
  const bool rel = a->window >= 0;
  if (!q || !k || !v || !o || !d_o || !lse || !dq || !dk || !dv || !delta_ws) return EVT_EINVAL;
  if (rel && (!emb_k || !emb_v || !demb_k || !demb_v)) return EVT_EINVAL;
  RP p = make_rp(a);
  p.q = q; p.k = k; p.v = v; p.o = o; p.d_o = d_o; p.ek = emb_k; p.ev = emb_v; p.lens_q = lens_q; p.lens_k = lens_k;
  p.lse = const_cast<float*>(lse);
  p.dq = dq; p.dk = dk; p.dv = dv; p.dek = demb_k; p.dev = demb_v; p.delta = delta_ws;
  hipStream_t st = (hipStream_t)stream;
  if (a->dtype == EVT_DT_F32) {
    evt_set_last_tag("mha_bwd_f32<%d>", (int)rel);
    rc = rel ? launch_bwd_f32<true>(p, st) : launch_bwd_f32<false>(p, st);
  } else {
    evt_set_last_tag("mha_bwd_bf16<%d, %d>", a->D / 32, (int)rel);
#define MHA_B(DK) rc = rel ? launch_bwd_bf16<DK, true>(p, st) : launch_bwd_bf16<DK, false>(p, st)
    switch (a->D / 32) { case 1: MHA_B(1); break; case 2: MHA_B(2); break; case 3: MHA_B(3); break; default: MHA_B(4); break; }
#undef MHA_B
  }
  return rc ? rc : evt_check_launch();
}

}  // extern "C"
