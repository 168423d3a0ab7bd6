// Fused windowed-relative-position self-attention of the s2 encoders (enc_p), gfx950, bf16 MFMA.
//
// Reference: src/easevoice/module/attentions.py:214-292 (MultiHeadAttention.attention with window_size = 4):
//   scores = (q/sqrt(d)) k^T + rel_to_abs((q/sqrt(d)) Ek^T);  masked_fill(mask == 0, -1e4);  p = drop(softmax(scores));
//   out    = p v + abs_to_rel(p) Ev
// which the reference (and the torch path this replaces) runs as ~18 launches forward and ~40 backward per layer over
// [B, 2, T, T] tensors (two bmm, two skew pad/reshape chains, masked_fill, softmax, dropout, two more bmm ...).
// Here: one forward launch, two backward launches, no [T, T] tensor in HBM.
//
// Flash structure with the band folded in.  For one (batch, head) a wave owns 16 queries; scores are produced as
// S^T = K Q^T (rows = keys, cols = queries) so the lane that owns query n keeps its softmax statistics, and P^T is
// directly the B operand of O^T += V^T P^T (A operand through ds_read_b64_tr_b16).  The relative part:
//   * logits: qe[i][r] = q_i . Ek[r] is ONE extra 16x16 MFMA tile per query tile (rows = the 2w+1 <= 16 offsets), parked
//     in LDS and added to the scores of keys j = i + r - w;
//   * values: the (at most 2w+1) probabilities of a query that sit on the band are recorded as raw scores while the key
//     loop runs, turned into probabilities with the final max / sum, and applied as one more MFMA against Ev.
// Key padding: keys j >= lens[b] are excluded (the reference's -1e4 fill underflows to exactly 0 in fp32 whenever the
// row has a live key); query rows i >= lens[b] are written as zeros (see enc_ops.hip for why that is equivalent).
// Dropout: keep(b, h, i, j) = hash(*seed_dev, site, ...) as in enc_ops.hip -- regenerated in both backward kernels.
//
// Backward: relattn_bwd_dq (same ownership as forward: dQ, and the 2 x [2w+1][D] embedding gradients) and
// relattn_bwd_dkv (a wave owns 16 keys, S = Q K^T layout: dK, dV), both recompute P from the saved log-sum-exp.
#include "evt_common.h"
#include "../../include/evt.h"

namespace {

struct RP {
  const bf16_t* q; const bf16_t* k; const bf16_t* v;     // [B][T][ld] rows, head h at column h*D
  const bf16_t* o; const bf16_t* d_o;                       // [B][T][ldo]
  bf16_t* out; bf16_t* dq; bf16_t* dk; bf16_t* dv;
  const float* ek; const float* ev;                         // [Hr][R][D] fp32 parameters
  float* dek; float* dev;                                   // fp32 accumulators (+=)
  const int* lens;                                          // [B] or null
  float* lse; float* delta;                                 // [B*H][T]
  int B, T, H, Hr, R, w;
  long ld, ldo;
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
__device__ __forceinline__ void stage_rows(bf16_t* dst, const bf16_t* src, long ld, int row0, int nrows, int limit) {
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
__device__ __forceinline__ void stage_emb(bf16_t* dst, const float* src, int R, int nrows) {
  for (int i = threadIdx.x; i < nrows * D; i += 256) {
    const int r = i / D, c = i - r * D;
    dst[r * PITCH + c] = r < R ? f2bf(src[r * D + c]) : (bf16_t)0;
  }
}

// ---------------------------------------------------------------------------------------------------------
// forward: block = 64 queries of one (b, h), 4 waves x 16 queries; keys in tiles of 32
// ---------------------------------------------------------------------------------------------------------
template <int DK>
__global__ __launch_bounds__(256) void relattn_fwd(RP p) {
  constexpr int D = 32 * DK, PITCH = D + 8, NDT = D / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* Qs = reinterpret_cast<bf16_t*>(smem);            // [64][PITCH]
  bf16_t* Ks = Qs + 64 * PITCH;                            // [32][PITCH]
  bf16_t* Vs = Ks + 32 * PITCH;                            // [32][PITCH]
  bf16_t* Eks = Vs + 32 * PITCH;                           // [16][PITCH]
  bf16_t* Evs = Eks + 16 * PITCH;                          // [32][PITCH]
  float* qe_l = reinterpret_cast<float*>(Evs + 32 * PITCH);   // [4][16][RB]
  float* sb_l = qe_l + 4 * 16 * RB;                        // [4][16][RB] raw band scores
  bf16_t* rw_l = reinterpret_cast<bf16_t*>(sb_l + 4 * 16 * RB);   // [4][16][40]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
  const int qb0 = blockIdx.x * 64;
  const int len = p.lens ? min(p.lens[b], p.T) : p.T;
  bf16_t* O = p.out + (long)b * p.T * p.ldo + h * D;
  float* LSE = p.lse + (long)bh * p.T;
  if (qb0 >= len) {   // block of padded queries: zeros
    for (int i = tid; i < 64 * (D / 8); i += 256) {
      const int r = i / (D / 8), c8 = i - r * (D / 8);
      if (qb0 + r < p.T) *reinterpret_cast<uint4*>(O + (long)(qb0 + r) * p.ldo + c8 * 8) = make_uint4(0, 0, 0, 0);
    }
    if (tid < 64 && qb0 + tid < p.T) LSE[qb0 + tid] = 0.f;
    return;
  }
  const bf16_t* Q = p.q + (long)b * p.T * p.ld + h * D;
  const bf16_t* K = p.k + (long)b * p.T * p.ld + h * D;
  const bf16_t* V = p.v + (long)b * p.T * p.ld + h * D;
  const int hr = h % p.Hr;
  stage_rows<D, PITCH>(Qs, Q, p.ld, qb0, 64, len);
  stage_emb<D, PITCH>(Eks, p.ek + (long)hr * p.R * D, p.R, 16);
  stage_emb<D, PITCH>(Evs, p.ev + (long)hr * p.R * D, p.R, 32);
  __syncthreads();

  const int q0 = qb0 + wave * 16, qi = q0 + n;
  bf16x8 qf[DK];
#pragma unroll
  for (int s = 0; s < DK; ++s) qf[s] = ld8(Qs + (wave * 16 + n) * PITCH + s * 32 + g * 8);
  float* qe = qe_l + wave * 16 * RB;
  float* sb = sb_l + wave * 16 * RB;
  {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < DK; ++s)
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld8(Eks + n * PITCH + s * 32 + g * 8), qf[s], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) { qe[n * RB + g * 4 + r] = acc[r] * p.scale; sb[n * RB + g * 4 + r] = -INFINITY; }
  }
  const unsigned drow = drop_row(drop_key(p), (unsigned)bh, qi);
  float m_run = -INFINITY, l_run = 0.f;
  f32x4 ot[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) ot[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < len; k0 += 32) {
    __syncthreads();
    stage_rows<D, PITCH>(Ks, K, p.ld, k0, 32, len);
    stage_rows<D, PITCH>(Vs, V, p.ld, k0, 32, len);
    __syncthreads();
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < DK; ++s) {
      s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld8(Ks + n * PITCH + s * 32 + g * 8), qf[s], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld8(Ks + (16 + n) * PITCH + s * 32 + g * 8), qf[s], s1, 0, 0, 0);
    }
    const bool band = (k0 <= q0 + 15 + p.w) && (k0 + 31 >= q0 - p.w);   // wave-uniform
    float sc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int kj = k0 + slot32(g, e);
      float v = (e < 4 ? s0[e] : s1[e - 4]) * p.scale;
      if (band) {
        const int rr = kj - qi + p.w;
        if (rr >= 0 && rr < p.R) {
          v += qe[n * RB + rr];
          if (kj < len) sb[n * RB + rr] = v;
        }
      }
      sc[e] = kj < len ? v : -INFINITY;
    }
    float mx = fmaxf(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])), fmaxf(fmaxf(sc[4], sc[5]), fmaxf(sc[6], sc[7])));
    mx = quad_max(mx);
    const float mn = fmaxf(m_run, mx);       // finite: tile k0 = 0 always holds key 0 < len
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
    const bf16x8 pf = pack8(sc);
    const bf16_t* vrow = Vs + (g * 4 + (n >> 2)) * PITCH + 4 * (n & 3);
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) ot[dt][r] *= alpha;
      ot[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr2(vrow + dt * 16, vrow + 16 * PITCH + dt * 16), pf, ot[dt], 0, 0, 0);
    }
  }
  const float inv = 1.f / l_run;
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
    for (int r = 0; r < 4; ++r) ot[dt][r] *= inv;
  // relative values: the band probabilities with the final statistics, one MFMA step against Ev
  __syncthreads();
  bf16_t* rw = rw_l + wave * 16 * 40;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int rr = g * 4 + r;
    const float s = sb[n * RB + rr];
    float pv = 0.f;
    if (rr < p.R && s > -INFINITY) pv = __expf(s - m_run) * inv * drop_mult(p, drow, qi + rr - p.w);
    rw[n * 40 + rr] = f2bf(pv);
    rw[n * 40 + 16 + rr] = (bf16_t)0;
  }
  __syncthreads();
  {
    const bf16x8 rf = ld8(rw + n * 40 + g * 8);
    const bf16_t* erow = Evs + (g * 8 + (n >> 2)) * PITCH + 4 * (n & 3);
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
      ot[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr2(erow + dt * 16, erow + 4 * PITCH + dt * 16), rf, ot[dt], 0, 0, 0);
  }
  if (qi < p.T) {
    const bool live = qi < len;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      bf16_t o4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) o4[r] = live ? f2bf(ot[dt][r]) : (bf16_t)0;
      *reinterpret_cast<uint2*>(O + (long)qi * p.ldo + dt * 16 + g * 4) = *reinterpret_cast<uint2*>(o4);
    }
    if (g == 0) LSE[qi] = live ? m_run + __logf(l_run) : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------------------
// dQ + embedding gradients: same ownership as forward
// ---------------------------------------------------------------------------------------------------------
template <int DK>
__global__ __launch_bounds__(256) void relattn_bwd_dq(RP p) {
  constexpr int D = 32 * DK, PITCH = D + 8, NDT = D / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* Qs = reinterpret_cast<bf16_t*>(smem);            // [64][PITCH]
  bf16_t* dOs = Qs + 64 * PITCH;                           // [64][PITCH]
  bf16_t* Ks = dOs + 64 * PITCH;                           // [32][PITCH]
  bf16_t* Vs = Ks + 32 * PITCH;                            // [32][PITCH]
  bf16_t* Eks = Vs + 32 * PITCH;                           // [32][PITCH] (rows >= R zero)
  bf16_t* Evs = Eks + 32 * PITCH;                          // [16][PITCH]
  float* qe_l = reinterpret_cast<float*>(Evs + 16 * PITCH);   // [64][RB]
  float* de_l = qe_l + 64 * RB;                            // [64][RB]  dO . Ev[r]
  float* ds_l = de_l + 64 * RB;                            // [64][RB]  dS on the band
  float* pb_l = ds_l + 64 * RB;                            // [64][RB]  dropped P on the band
  float* dl_l = pb_l + 64 * RB;                            // [64] delta
  bf16_t* dw_l = reinterpret_cast<bf16_t*>(dl_l + 64);     // [4][16][40]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
  const int qb0 = blockIdx.x * 64;
  const int len = p.lens ? min(p.lens[b], p.T) : p.T;
  bf16_t* dQ = p.dq + (long)b * p.T * p.ld + h * D;
  float* DL = p.delta + (long)bh * p.T;
  if (qb0 >= len) {
    for (int i = tid; i < 64 * (D / 8); i += 256) {
      const int r = i / (D / 8), c8 = i - r * (D / 8);
      if (qb0 + r < p.T) *reinterpret_cast<uint4*>(dQ + (long)(qb0 + r) * p.ld + c8 * 8) = make_uint4(0, 0, 0, 0);
    }
    if (tid < 64 && qb0 + tid < p.T) DL[qb0 + tid] = 0.f;
    return;
  }
  const bf16_t* Q = p.q + (long)b * p.T * p.ld + h * D;
  const bf16_t* K = p.k + (long)b * p.T * p.ld + h * D;
  const bf16_t* V = p.v + (long)b * p.T * p.ld + h * D;
  const bf16_t* Og = p.o + (long)b * p.T * p.ldo + h * D;
  const bf16_t* dOg = p.d_o + (long)b * p.T * p.ldo + h * D;
  const int hr = h % p.Hr;
  stage_rows<D, PITCH>(Qs, Q, p.ld, qb0, 64, len);
  stage_rows<D, PITCH>(dOs, dOg, p.ldo, qb0, 64, len);
  stage_emb<D, PITCH>(Eks, p.ek + (long)hr * p.R * D, p.R, 32);
  stage_emb<D, PITCH>(Evs, p.ev + (long)hr * p.R * D, p.R, 16);
  for (int i = tid; i < 64 * RB; i += 256) { ds_l[i] = 0.f; pb_l[i] = 0.f; }
  if (tid < 64) dl_l[tid] = 0.f;
  __syncthreads();
  // delta_i = dO_i . O_i : 4 lanes per row, 16-byte pieces
  {
    const int r = tid >> 2, part = tid & 3;
    float acc = 0.f;
    if (qb0 + r < len) {
      for (int c8 = part; c8 < D / 8; c8 += 4) {
        const uint4 ov = *reinterpret_cast<const uint4*>(Og + (long)(qb0 + r) * p.ldo + c8 * 8);
        const bf16_t* po = reinterpret_cast<const bf16_t*>(&ov);
        const bf16_t* pd = dOs + r * PITCH + c8 * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += bf2f(po[e]) * bf2f(pd[e]);
      }
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    if (part == 0) { dl_l[r] = acc; if (qb0 + r < p.T) DL[qb0 + r] = acc; }
  }
  const int q0 = qb0 + wave * 16, qi = q0 + n;
  const int ql = wave * 16 + n;       // row inside the block tiles
  bf16x8 qf[DK], dof[DK];
#pragma unroll
  for (int s = 0; s < DK; ++s) {
    qf[s] = ld8(Qs + ql * PITCH + s * 32 + g * 8);
    dof[s] = ld8(dOs + ql * PITCH + s * 32 + g * 8);
  }
  {
    f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < DK; ++s) {
      a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld8(Eks + n * PITCH + s * 32 + g * 8), qf[s], a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld8(Evs + n * PITCH + s * 32 + g * 8), dof[s], a2, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { qe_l[ql * RB + g * 4 + r] = a1[r] * p.scale; de_l[ql * RB + g * 4 + r] = a2[r]; }
  }
  __syncthreads();
  const float lse = qi < len ? p.lse[(long)bh * p.T + qi] : 0.f;
  const float dlt = dl_l[ql];
  const unsigned drow = drop_row(drop_key(p), (unsigned)bh, qi);
  f32x4 dqt[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) dqt[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < len; k0 += 32) {
    __syncthreads();
    stage_rows<D, PITCH>(Ks, K, p.ld, k0, 32, len);
    stage_rows<D, PITCH>(Vs, V, p.ld, k0, 32, len);
    __syncthreads();
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f}, d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < DK; ++s) {
      s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld8(Ks + n * PITCH + s * 32 + g * 8), qf[s], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld8(Ks + (16 + n) * PITCH + s * 32 + g * 8), qf[s], s1, 0, 0, 0);
      d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld8(Vs + n * PITCH + s * 32 + g * 8), dof[s], d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld8(Vs + (16 + n) * PITCH + s * 32 + g * 8), dof[s], d1, 0, 0, 0);
    }
    const bool band = (k0 <= q0 + 15 + p.w) && (k0 + 31 >= q0 - p.w);
    float ds[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int kj = k0 + slot32(g, e);
      float v = (e < 4 ? s0[e] : s1[e - 4]) * p.scale;
      float dp = (e < 4 ? d0[e] : d1[e - 4]);
      const int rr = kj - qi + p.w;
      const bool onb = band && rr >= 0 && rr < p.R;
      if (onb) { v += qe_l[ql * RB + rr]; dp += de_l[ql * RB + rr]; }
      const bool ok = kj < len && qi < len;
      const float pr = ok ? __expf(v - lse) : 0.f;
      const float mult = drop_mult(p, drow, kj);
      const float dsv = pr * (dp * mult - dlt);
      if (onb && ok) { ds_l[ql * RB + rr] = dsv; pb_l[ql * RB + rr] = pr * mult; }
      ds[e] = dsv;
    }
    const bf16x8 dsf = pack8(ds);
    const bf16_t* krow = Ks + (g * 4 + (n >> 2)) * PITCH + 4 * (n & 3);
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
      dqt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr2(krow + dt * 16, krow + 16 * PITCH + dt * 16), dsf, dqt[dt], 0, 0, 0);
  }
  // band part of dQ: dS[i, i+r-w] * Ek[r]
  __syncthreads();
  bf16_t* dw = dw_l + wave * 16 * 40;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    dw[n * 40 + g * 4 + r] = f2bf(ds_l[ql * RB + g * 4 + r]);
    dw[n * 40 + 16 + g * 4 + r] = (bf16_t)0;
  }
  __syncthreads();
  {
    const bf16x8 rf = ld8(dw + n * 40 + g * 8);
    const bf16_t* erow = Eks + (g * 8 + (n >> 2)) * PITCH + 4 * (n & 3);
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
      dqt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr2(erow + dt * 16, erow + 4 * PITCH + dt * 16), rf, dqt[dt], 0, 0, 0);
  }
  if (qi < p.T) {
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      bf16_t o4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) o4[r] = qi < len ? f2bf(dqt[dt][r] * p.scale) : (bf16_t)0;
      *reinterpret_cast<uint2*>(dQ + (long)qi * p.ld + dt * 16 + g * 4) = *reinterpret_cast<uint2*>(o4);
    }
  }
  // embedding gradients of this block: dEk[r][d] += scale * sum_q dS_band[q][r] Q[q][d];  dEv[r][d] += sum_q P_band[q][r] dO[q][d]
  float* dek = p.dek + (long)hr * p.R * D;
  float* dev = p.dev + (long)hr * p.R * D;
  for (int i = tid; i < p.R * D; i += 256) {
    const int r = i / D, c = i - r * D;
    float a1 = 0.f, a2 = 0.f;
    for (int q = 0; q < 64; ++q) {
      a1 += ds_l[q * RB + r] * bf2f(Qs[q * PITCH + c]);
      a2 += pb_l[q * RB + r] * bf2f(dOs[q * PITCH + c]);
    }
    atomicAdd(dek + i, a1 * p.scale);
    atomicAdd(dev + i, a2);
  }
}

// ---------------------------------------------------------------------------------------------------------
// dK, dV: block = 64 keys of one (b, h), a wave owns 16 keys; S = Q K^T layout (rows = queries, cols = keys);
// queries in tiles of 32.  P / dS are directly the B operands of dV^T += dO^T P and dK^T += Q^T dS.
// ---------------------------------------------------------------------------------------------------------
template <int DK>
__global__ __launch_bounds__(256) void relattn_bwd_dkv(RP p) {
  constexpr int D = 32 * DK, PITCH = D + 8, NDT = D / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);            // [64][PITCH]
  bf16_t* Vs = Ks + 64 * PITCH;                            // [64][PITCH]
  bf16_t* Qs = Vs + 64 * PITCH;                            // [32][PITCH]
  bf16_t* dOs = Qs + 32 * PITCH;                           // [32][PITCH]
  bf16_t* Eks = dOs + 32 * PITCH;                          // [16][PITCH]
  bf16_t* Evs = Eks + 16 * PITCH;                          // [16][PITCH]
  float* qe_l = reinterpret_cast<float*>(Evs + 16 * PITCH);   // [32][RB]
  float* de_l = qe_l + 32 * RB;                            // [32][RB]
  float* ls_l = de_l + 32 * RB;                            // [32] lse
  float* dl_l = ls_l + 32;                                 // [32] delta

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
  const int kb0 = blockIdx.x * 64;
  const int len = p.lens ? min(p.lens[b], p.T) : p.T;
  bf16_t* dKg = p.dk + (long)b * p.T * p.ld + h * D;
  bf16_t* dVg = p.dv + (long)b * p.T * p.ld + h * D;
  if (kb0 >= len) {
    for (int i = tid; i < 64 * (D / 8); i += 256) {
      const int r = i / (D / 8), c8 = i - r * (D / 8);
      if (kb0 + r < p.T) {
        *reinterpret_cast<uint4*>(dKg + (long)(kb0 + r) * p.ld + c8 * 8) = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(dVg + (long)(kb0 + r) * p.ld + c8 * 8) = make_uint4(0, 0, 0, 0);
      }
    }
    return;
  }
  const bf16_t* Q = p.q + (long)b * p.T * p.ld + h * D;
  const bf16_t* K = p.k + (long)b * p.T * p.ld + h * D;
  const bf16_t* V = p.v + (long)b * p.T * p.ld + h * D;
  const bf16_t* dOg = p.d_o + (long)b * p.T * p.ldo + h * D;
  const int hr = h % p.Hr;
  stage_rows<D, PITCH>(Ks, K, p.ld, kb0, 64, len);
  stage_rows<D, PITCH>(Vs, V, p.ld, kb0, 64, len);
  stage_emb<D, PITCH>(Eks, p.ek + (long)hr * p.R * D, p.R, 16);
  stage_emb<D, PITCH>(Evs, p.ev + (long)hr * p.R * D, p.R, 16);
  __syncthreads();
  const int k0w = kb0 + wave * 16, kj = k0w + n;     // my key (MFMA column)
  bf16x8 kf[DK], vf[DK];
#pragma unroll
  for (int s = 0; s < DK; ++s) {
    kf[s] = ld8(Ks + (wave * 16 + n) * PITCH + s * 32 + g * 8);
    vf[s] = ld8(Vs + (wave * 16 + n) * PITCH + s * 32 + g * 8);
  }
  const unsigned dkey = drop_key(p);
  f32x4 dkt[NDT], dvt[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) { dkt[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dvt[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  for (int q0 = 0; q0 < len; q0 += 32) {
    __syncthreads();
    stage_rows<D, PITCH>(Qs, Q, p.ld, q0, 32, len);
    stage_rows<D, PITCH>(dOs, dOg, p.ldo, q0, 32, len);
    if (tid < 32) {
      const bool ok = q0 + tid < len;
      ls_l[tid] = ok ? p.lse[(long)bh * p.T + q0 + tid] : 0.f;
      dl_l[tid] = ok ? p.delta[(long)bh * p.T + q0 + tid] : 0.f;
    }
    __syncthreads();
    // band tables of this query tile (block-uniform test): waves 0/1 -> q.Ek, waves 2/3 -> dO.Ev, 16 queries each
    const bool band = (kb0 <= q0 + 31 + p.w) && (kb0 + 63 >= q0 - p.w);
    if (band) {
      const int tl = wave & 1;
      const bf16_t* As = (wave < 2 ? Qs : dOs) + (tl * 16 + n) * PITCH + g * 8;      // A: m = query
      const bf16_t* Bs = (wave < 2 ? Eks : Evs) + n * PITCH + g * 8;                  // B: n = offset r
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < DK; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld8(As + s * 32), ld8(Bs + s * 32), acc, 0, 0, 0);
      float* dst = wave < 2 ? qe_l : de_l;          // result: lane (n = r, g) holds query tl*16 + g*4 + rr
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[(tl * 16 + g * 4 + r) * RB + n] = wave < 2 ? acc[r] * p.scale : acc[r];
    }
    __syncthreads();
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f}, d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < DK; ++s) {
      s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld8(Qs + n * PITCH + s * 32 + g * 8), kf[s], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld8(Qs + (16 + n) * PITCH + s * 32 + g * 8), kf[s], s1, 0, 0, 0);
      d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld8(dOs + n * PITCH + s * 32 + g * 8), vf[s], d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld8(dOs + (16 + n) * PITCH + s * 32 + g * 8), vf[s], d1, 0, 0, 0);
    }
    float pd[8], ds[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int qloc = slot32(g, e), qi = q0 + qloc;
      float v = (e < 4 ? s0[e] : s1[e - 4]) * p.scale;
      float dp = (e < 4 ? d0[e] : d1[e - 4]);
      const int rr = kj - qi + p.w;
      if (band && rr >= 0 && rr < p.R) { v += qe_l[qloc * RB + rr]; dp += de_l[qloc * RB + rr]; }
      const bool ok = kj < len && qi < len;
      const float pr = ok ? __expf(v - ls_l[qloc]) : 0.f;
      const float mult = drop_mult(p, drop_row(dkey, (unsigned)bh, qi), kj);
      pd[e] = pr * mult;
      ds[e] = pr * (dp * mult - dl_l[qloc]);
    }
    const bf16x8 pf = pack8(pd), dsf = pack8(ds);
    const bf16_t* orow = dOs + (g * 4 + (n >> 2)) * PITCH + 4 * (n & 3);
    const bf16_t* qrow = Qs + (g * 4 + (n >> 2)) * PITCH + 4 * (n & 3);
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      dvt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr2(orow + dt * 16, orow + 16 * PITCH + dt * 16), pf, dvt[dt], 0, 0, 0);
      dkt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr2(qrow + dt * 16, qrow + 16 * PITCH + dt * 16), dsf, dkt[dt], 0, 0, 0);
    }
  }
  if (kj < p.T) {
    const bool live = kj < len;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      bf16_t k4[4], v4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        k4[r] = live ? f2bf(dkt[dt][r] * p.scale) : (bf16_t)0;
        v4[r] = live ? f2bf(dvt[dt][r]) : (bf16_t)0;
      }
      *reinterpret_cast<uint2*>(dKg + (long)kj * p.ld + dt * 16 + g * 4) = *reinterpret_cast<uint2*>(k4);
      *reinterpret_cast<uint2*>(dVg + (long)kj * p.ld + dt * 16 + g * 4) = *reinterpret_cast<uint2*>(v4);
    }
  }
}

template <int DK> constexpr size_t fwd_lds() {
  constexpr int PITCH = 32 * DK + 8;
  return (size_t)(64 + 32 + 32 + 16 + 32) * PITCH * 2 + 2 * 4 * 16 * RB * 4 + 4 * 16 * 40 * 2;
}
template <int DK> constexpr size_t dq_lds() {
  constexpr int PITCH = 32 * DK + 8;
  return (size_t)(64 + 64 + 32 + 32 + 32 + 16) * PITCH * 2 + (4 * 64 * RB + 64) * 4 + 4 * 16 * 40 * 2;
}
template <int DK> constexpr size_t dkv_lds() {
  constexpr int PITCH = 32 * DK + 8;
  return (size_t)(64 + 64 + 32 + 32 + 16 + 16) * PITCH * 2 + (2 * 32 * RB + 64) * 4;
}

int check(const evt_relattn_params* a) {
  if (!a || a->B <= 0 || a->T <= 0 || a->H <= 0 || a->D <= 0 || a->window < 0) return EVT_EINVAL;
  if (a->D % 32 || a->D > 128) return EVT_ENOTSUP;
  if (2 * a->window + 1 > 16) return EVT_ENOTSUP;
  if (a->n_heads_rel <= 0 || a->ld < (int64_t)a->H * a->D || a->ldo < (int64_t)a->H * a->D) return EVT_EINVAL;
  if (a->ld % 8 || a->ldo % 8) return EVT_EINVAL;            // 16-byte row pieces
  if (a->dropout_p < 0.f || a->dropout_p >= 1.f) return EVT_EINVAL;
  return EVT_OK;
}

RP make_rp(const evt_relattn_params* a) {
  RP p{};
  p.B = a->B; p.T = a->T; p.H = a->H; p.Hr = a->n_heads_rel; p.w = a->window; p.R = 2 * a->window + 1;
  p.ld = a->ld; p.ldo = a->ldo;
  p.scale = 1.f / sqrtf((float)a->D);
  p.thr = a->dropout_p > 0.f ? (unsigned)fminf(a->dropout_p * 4294967296.f, 4294967040.f) : 0u;
  p.keep_scale = a->dropout_p > 0.f ? 1.f / (1.f - a->dropout_p) : 1.f;
  p.seed_dev = a->seed_dev; p.site = a->site;
  return p;
}

template <typename F>
int set_lds(F fn, size_t lds) {
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return EVT_ELAUNCH;
  return EVT_OK;
}

}  // namespace

extern "C" {

int evt_relattn_fwd(const evt_relattn_params* a, const void* q, const void* k, const void* v, const float* emb_k,
                    const float* emb_v, const int32_t* lens, void* out, float* lse, void* stream) {
  int rc = check(a);
  if (rc) return rc;
  if (!q || !k || !v || !emb_k || !emb_v || !out || !lse) return EVT_EINVAL;
  RP p = make_rp(a);
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.ek = emb_k; p.ev = emb_v; p.lens = lens;
  p.out = (bf16_t*)out; p.lse = lse;
  const dim3 grid((a->T + 63) / 64, a->B * a->H);
  hipStream_t st = (hipStream_t)stream;
  evt_set_last_tag("relattn_fwd<%d>", a->D);
#define RA_FWD(DK) { static bool once = false; if (!once) { if (set_lds(&relattn_fwd<DK>, fwd_lds<DK>())) return EVT_ELAUNCH; once = true; } \
                     hipLaunchKernelGGL(relattn_fwd<DK>, grid, dim3(256), fwd_lds<DK>(), st, p); }
  switch (a->D / 32) { case 1: RA_FWD(1) break; case 2: RA_FWD(2) break; case 3: RA_FWD(3) break; default: RA_FWD(4) break; }
#undef RA_FWD
  return evt_check_launch();
}

int evt_relattn_bwd(const evt_relattn_params* a, const void* q, const void* k, const void* v, const void* o,
                    const void* d_o, const float* lse, const float* emb_k, const float* emb_v, const int32_t* lens,
                    void* dq, void* dk, void* dv, float* demb_k, float* demb_v, float* delta_ws, void* stream) {
  int rc = check(a);
  if (rc) return rc;
  if (!q || !k || !v || !o || !d_o || !lse || !emb_k || !emb_v || !dq || !dk || !dv || !demb_k || !demb_v || !delta_ws)
    return EVT_EINVAL;
  RP p = make_rp(a);
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = (const bf16_t*)o;
  p.d_o = (const bf16_t*)d_o; p.ek = emb_k; p.ev = emb_v; p.lens = lens; p.lse = const_cast<float*>(lse);
  p.dq = (bf16_t*)dq; p.dk = (bf16_t*)dk; p.dv = (bf16_t*)dv; p.dek = demb_k; p.dev = demb_v; p.delta = delta_ws;
  const dim3 grid((a->T + 63) / 64, a->B * a->H);
  hipStream_t st = (hipStream_t)stream;
  evt_set_last_tag("relattn_bwd<%d>", a->D);
#define RA_BWD(DK) { static bool once = false; if (!once) { if (set_lds(&relattn_bwd_dq<DK>, dq_lds<DK>()) || set_lds(&relattn_bwd_dkv<DK>, dkv_lds<DK>())) return EVT_ELAUNCH; once = true; } \
                     hipLaunchKernelGGL(relattn_bwd_dq<DK>, grid, dim3(256), dq_lds<DK>(), st, p);                          \
                     hipLaunchKernelGGL(relattn_bwd_dkv<DK>, grid, dim3(256), dkv_lds<DK>(), st, p); }
  switch (a->D / 32) { case 1: RA_BWD(1) break; case 2: RA_BWD(2) break; case 3: RA_BWD(3) break; default: RA_BWD(4) break; }
#undef RA_BWD
  return evt_check_launch();
}

}  // extern "C"
