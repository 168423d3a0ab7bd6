// resunit_bwd: the whole backward of one HiFi-GAN ResBlock1 step  y = x + c2(lrelu(c1(lrelu(x))))  (reference:
// src/easevoice/module/modules.py:299-308 through torch.autograd) as ONE launch for the narrow vocoder stages (C = 16 at
// 20480 samples, C = 32 at 10240), gfx950 bf16: both backward-data convolutions, both weight gradients and both bias
// gradients.  Unfused this is four launches per step (two backward-data, two weight gradients: 70 - 100 us for 42 MB of
// tensor traffic); here dy, xa = lrelu(x) and mid_a = lrelu(c1(xa)) are read once and dx is written once.
//
//   dmid = (c2^T dy) * lrelu'(mid_a)        on the positions c1^T needs: [q0 - h1, q0 + 63 + h1]
//   dx   = (c1^T dmid) * lrelu'(xa) + dy    on the unit's own 64 positions
//   dW2[co][t][ci] = sum_q dy[q - t + h2][co]     * mid_a[q][ci]      q over the unit's own positions
//   dW1[co][t][ci] = sum_q dmid[q - t*d + h1][co] * xa[q][ci]
//   db2 = sum_q dy[q],  db1 = sum_q dmid[q]
// (the weight gradients are written with the roles of the shifted operand swapped against the textbook form
// sum_q dy[q] mid_a[q + t - h2]: every pair of positions still belongs to exactly one unit -- the one that owns the
// un-shifted row -- and the shifted operand is the one whose halo this unit holds anyway).
//
// Structure = resunit_fwd's: waves are independent and persistent over 64-position units, the two ALT weight images
// (tap-flipped transposes: backward-data of a "same" convolution is the same convolution with them) sit in LDS for the
// life of the block, each wave stages the rows of its unit in a private LDS area:
//     DY  [64 + 2(h1+h2) (+1 padded tap)]   dy rows (scaled by dy_scale on the way in: the stage mean's 1/3)
//     MID [64 + 2 h1]                       mid_a rows (gate of dmid; the own 64 are the weight gradient's operand)
//     XA  [64]                              xa rows (gate of dx, weight-gradient operand)
//     DM  [16 NT1]                          dmid, produced by stage A, consumed by stage B and by dW1
// Stage A / B are resunit_fwd's two convolutions in reverse order (first the undilated one on 64 + 2 h1 positions, then
// the dilated one on 64).  The weight gradients are MFMAs with M = dy channel, N = x channel, K = positions: both
// operands are position-major in LDS, fragments come from ds_read_b64_tr_b16; a lane group reads rows
// {4g..4g+3} and {16+4g..16+4g+3} of a 32-position block (the same permutation of K for both operands), which makes the
// eight rows of a read tile one 256-byte bank window for 32-byte rows (C = 16) and, with the slot swizzle below, for
// 64-byte rows (C = 32).  The accumulators ([C][taps][C] fp32 per convolution: 88 registers at C = 16, 352 at C = 32,
// where LDS allows one wave per SIMD anyway) live in registers across the persistent loop; at the end the four waves
// add theirs in LDS in wave order and the block stores ONE partial row; fold_partials_multi adds the rows of all blocks
// in block order into the gradient images / bias gradients: bit-identical from run to run.
//
// LDS rows are 2C bytes, no padding.  C = 32: the 16-byte slot s of row r lives at slot s ^ (2 * ((r >> 2) & 1)) --
// conflict-free for the 16-byte MFMA-operand reads (rows n + shift, any shift) and for the transpose reads.
#include "resunit_common.h"
#include "../../include/evt.h"
#include <cstdlib>

namespace {

using namespace evt_ru;

struct RBP {
  const h16_t* dy; const h16_t* xa; const h16_t* mid; const h16_t* w1; const h16_t* w2;   // w: ALT images
  h16_t* dx; h16_t* dmid;          // dmid: optional copy of the unit's own rows of dmid (null: not written)
  float* part; long part_stride;     // partial rows (one per block) or null: no weight gradients
  int nseq, L, dil;
  float slope, dy_scale;
  int ups; long total;
  int h1;
  int dyrows, midrows;               // rows staged per unit
  int r_mid, r_xa, r_dm, wave_rows;  // region starts (rows) inside a wave's LDS area, rows per wave
};

// weight gradient of one convolution over the unit's own 64 positions (two K steps of 32): acc[a][t][b] += A(t) x X with
//   X = own rows of the un-shifted operand: region xreg, row xbase + o;
//   A(t) = rows of the shifted operand: region areg, row abase - t * tstep + o.
// The fragments of step (t, ks) are requested D steps ahead into a rotating buffer; bs[a] collects the column sums of the
// un-shifted A rows (centre tap) = the bias gradient.
template <int CI, int KR, int MT>
__device__ __forceinline__ void wgrad_conv(f32x4 (&acc)[MT][KR][MT], float (&bs)[MT], const unsigned char* xreg, const int xbase,
                                           const unsigned char* areg, const int abase, const int tstep, const int n, const int g) {
  constexpr int PITCH = CI * 2, S = 2 * KR, D = 4, H2 = (KR - 1) / 2;
  const int lrow = g * 4 + (n >> 2), colb = (n & 3) * 8;
  u32x4 bq[2][MT], fa[D][MT];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int b = 0; b < MT; ++b) bq[ks][b] = tr_frag<PITCH>(xreg + tr_off<CI>(xbase + ks * 32 + lrow, b, colb));
  auto issue = [&](const int s) {
    const int t = s >> 1, ks = s & 1;
    const int row = abase - t * tstep + ks * 32 + lrow;
#pragma unroll
    for (int a = 0; a < MT; ++a) fa[s % D][a] = tr_frag<PITCH>(areg + tr_off<CI>(row, a, colb));
  };
#pragma unroll
  for (int s = 0; s < D; ++s) issue(s);
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int b = 0; b < MT; ++b) tie(bq[ks][b]);
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int t = s >> 1, ks = s & 1;
#pragma unroll
    for (int a = 0; a < MT; ++a) tie(fa[s % D][a]);
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      if (t == H2) bs[a] += sum8(fa[s % D][a]);
#pragma unroll
      for (int b = 0; b < MT; ++b)
        acc[a][t][b] = EVT_MFMA_16x16x32(as_h8(fa[s % D][a]), as_h8(bq[ks][b]), acc[a][t][b], 0, 0, 0);
    }
    if (s + D < S) issue(s + D);
  }
}

// CI: channels; NK: K steps of 32 per convolution (padded taps * CI / 32); NT1: position tiles of stage A
// (16 * NT1 >= 64 + 2 h1); WG: weight gradients in the launch; PF: 0 = rows loaded at the top of a unit, 1 = next unit's
// rows prefetched into registers right after this unit's are published
template <int CI, int NK, int NT1, bool WG, int PF>
__device__ __forceinline__ void resunit_bwd_body(const RBP& p, unsigned char* smem, const int blk, const int nblk) {
  constexpr int MT = CI / 16;
  constexpr int PITCH = CI * 2;
  constexpr int LOGP = CI == 16 ? 1 : 2;
  constexpr int PPR = 1 << LOGP;                         // 16-byte pieces per row
  constexpr int KTOT = NK * 32;
  constexpr int KHP = KTOT / CI;                         // taps of the prepared images
  constexpr int KR = CI == 16 ? KHP - 1 : KHP;           // real taps (3 / 7 / 11)
  constexpr int H2 = (KR - 1) / 2;
  constexpr int WPITCH = KTOT * 2 + 16;
  constexpr int XDY = ((NT1 * 16 + KHP - 1) * PPR + 63) / 64;
  constexpr int XMID = (NT1 * 16 * PPR + 63) / 64;
  constexpr int XXA = PPR;
  constexpr int IMG = CI * KHP * CI;                     // floats of one gradient image
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  unsigned char* wl1 = smem;
  unsigned char* wl2 = smem + CI * WPITCH;
  unsigned char* wv = smem + 2 * CI * WPITCH + wave * p.wave_rows * PITCH;
  unsigned char* dyl = wv;
  unsigned char* midl = wv + p.r_mid * PITCH;
  unsigned char* xal = wv + p.r_xa * PITCH;
  unsigned char* dml = wv + p.r_dm * PITCH;

  // the wave's area starts as zeros (the rows of DM behind the last tile that a padded tap may read stay zero for good)
  for (int o = lane * 16; o < p.wave_rows * PITCH; o += 1024) *reinterpret_cast<uint4*>(wv + o) = make_uint4(0, 0, 0, 0);
  load_weights<CI, KTOT>(wl1, wl2, p.w1, p.w2);
  __syncthreads();

  const int h1 = p.h1;
  const long wave_id = (long)blk * 4 + wave, nwaves = (long)nblk * 4;
  const int ndy = p.dyrows * PPR, nmid = p.midrows * PPR;
  uint4 xdy[XDY], xmid[XMID], xxa[XXA];
  auto load_unit = [&](long u) {
    const int seq = (int)(u / p.ups);
    const int q0 = (int)(u - (long)seq * p.ups) * 64;
    const long sb = (long)seq * p.L * CI;
#pragma unroll
    for (int i = 0; i < XDY; ++i) {
      const int idx = lane + i * 64, r = idx >> LOGP, pc = idx & (PPR - 1);
      const int pos = q0 - h1 - H2 + r;
      xdy[i] = (idx < ndy && pos >= 0 && pos < p.L) ? *reinterpret_cast<const uint4*>(p.dy + sb + (long)pos * CI + pc * 8)
                                                     : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < XMID; ++i) {
      const int idx = lane + i * 64, r = idx >> LOGP, pc = idx & (PPR - 1);
      const int pos = q0 - h1 + r;
      xmid[i] = (idx < nmid && pos >= 0 && pos < p.L) ? *reinterpret_cast<const uint4*>(p.mid + sb + (long)pos * CI + pc * 8)
                                                       : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < XXA; ++i) {
      const int idx = lane + i * 64, r = idx >> LOGP, pc = idx & (PPR - 1);
      const int pos = q0 + r;
      xxa[i] = pos < p.L ? *reinterpret_cast<const uint4*>(p.xa + sb + (long)pos * CI + pc * 8) : make_uint4(0, 0, 0, 0);
    }
  };

  f32x4 acc2[MT][KR][MT], acc1[MT][KR][MT];              // weight gradients [dy tile][tap][x tile]
  float bs2[MT], bs1[MT];
  if constexpr (WG) {
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      bs2[a] = bs1[a] = 0.f;
#pragma unroll
      for (int t = 0; t < KR; ++t)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc2[a][t][b] = acc1[a][t][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }

  if (PF != 0 && wave_id < p.total) load_unit(wave_id);
  for (long u = wave_id; u < p.total; u += nwaves) {
    const int seq = (int)(u / p.ups);
    const int q0 = (int)(u - (long)seq * p.ups) * 64;
    const long sbase = (long)seq * p.L * CI;
    if constexpr (PF == 0) load_unit(u);
    // publish the unit's rows (the previous unit's reads are complete: in-order DS)
#pragma unroll
    for (int i = 0; i < XDY; ++i) {
      const int idx = lane + i * 64;
      if (idx < ndy)
        *reinterpret_cast<uint4*>(dyl + piece_off<CI>(idx >> LOGP, idx & (PPR - 1))) =
            p.dy_scale == 1.f ? xdy[i] : scale8(xdy[i], p.dy_scale);
    }
#pragma unroll
    for (int i = 0; i < XMID; ++i) {
      const int idx = lane + i * 64;
      if (idx < nmid) *reinterpret_cast<uint4*>(midl + piece_off<CI>(idx >> LOGP, idx & (PPR - 1))) = xmid[i];
    }
#pragma unroll
    for (int i = 0; i < XXA; ++i) {
      const int idx = lane + i * 64;
      *reinterpret_cast<uint4*>(xal + piece_off<CI>(idx >> LOGP, idx & (PPR - 1))) = xxa[i];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if constexpr (PF == 1)
      if (u + nwaves < p.total) load_unit(u + nwaves);

    // ---- stage A: dmid on DM rows m = 0 .. 16 NT1 - 1 (position q0 - h1 + m); DY row of (m, tap) = m + tap ----
    {
      f32x4 acc[MT][NT1];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT1; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      // the gates (mid_a rows) travel under the MFMAs
      u32x2 gv[NT1][MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const unsigned char* gb = midl + chan_off<CI>(n, i * 16 + g * 4);
#pragma unroll
        for (int j = 0; j < NT1; ++j) gv[j][i] = *reinterpret_cast<const u32x2*>(gb + j * 16 * PITCH);
      }
      conv_stage<CI, NK, MT, NT1>(acc, wl2, dyl, 1, n, g);
#pragma unroll
      for (int j = 0; j < NT1; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) tie(gv[j][i]);
      // gate by lrelu'(mid_a), zero outside the sequence and behind the last needed row; lane: channels i*16+g*4.., row j*16+n
#pragma unroll
      for (int j = 0; j < NT1; ++j) {
        const int m = j * 16 + n;
        const int pos = q0 - h1 + m;
        const bool inside = pos >= 0 && pos < p.L && m < 64 + 2 * h1;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const float g0 = h2f_lo(gv[j][i][0]), g1 = h2f_hi(gv[j][i][0]);
          const float g2 = h2f_lo(gv[j][i][1]), g3 = h2f_hi(gv[j][i][1]);
          const float gg[4] = {g0, g1, g2, g3};
          h16_t o4[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = acc[i][j][r] * (gg[r] > 0.f ? 1.f : p.slope);
            o4[r] = f2h(inside ? v : 0.f);
          }
          *reinterpret_cast<uint2*>(dml + chan_off<CI>(m, i * 16 + g * 4)) = *reinterpret_cast<uint2*>(o4);
          if (p.dmid && inside && m >= h1 && m < h1 + 64)
            *reinterpret_cast<uint2*>(p.dmid + sbase + (long)pos * CI + i * 16 + g * 4) = *reinterpret_cast<uint2*>(o4);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();

    // ---- stage B: dx on the own positions o = 0 .. 63; DM row of (o, tap) = o + tap * dil ----
    {
      f32x4 acc[MT][4];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      u32x2 gv[4][MT], dv[4][MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const unsigned char* gb = xal + chan_off<CI>(n, i * 16 + g * 4);
        const unsigned char* db = dyl + chan_off<CI>(h1 + H2 + n, i * 16 + g * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          gv[j][i] = *reinterpret_cast<const u32x2*>(gb + j * 16 * PITCH);
          dv[j][i] = *reinterpret_cast<const u32x2*>(db + j * 16 * PITCH);
        }
      }
      conv_stage<CI, NK, MT, 4>(acc, wl1, dml, p.dil, n, g);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) { tie(gv[j][i]); tie(dv[j][i]); }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int o = j * 16 + n;
        const int q = q0 + o;
        if (q >= p.L) continue;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const float gg[4] = {h2f_lo(gv[j][i][0]), h2f_hi(gv[j][i][0]),
                               h2f_lo(gv[j][i][1]), h2f_hi(gv[j][i][1])};
          const float dd[4] = {h2f_lo(dv[j][i][0]), h2f_hi(dv[j][i][0]),
                               h2f_lo(dv[j][i][1]), h2f_hi(dv[j][i][1])};
          h16_t o4[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) o4[r] = f2h(acc[i][j][r] * (gg[r] > 0.f ? 1.f : p.slope) + dd[r]);
          *reinterpret_cast<uint2*>(p.dx + sbase + (long)q * CI + i * 16 + g * 4) = *reinterpret_cast<uint2*>(o4);
        }
      }
    }

    // ---- weight gradients: K = the unit's own 64 positions ----
    if constexpr (WG) {
      // conv2: x operand = own rows of MID (row h1 + o); dy operand = DY row o + h1 + 2 H2 - t
      wgrad_conv<CI, KR, MT>(acc2, bs2, midl, h1, dyl, h1 + 2 * H2, 1, n, g);
      // conv1: x operand = XA row o; dmid operand = DM row o + 2 h1 - t * dil
      wgrad_conv<CI, KR, MT>(acc1, bs1, xal, 0, dml, 2 * h1, p.dil, n, g);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }

  // ---- the block's partial row: the four waves' accumulators added in wave order, one coalesced store ----
  if constexpr (WG) {
    if (p.part == nullptr) return;
    float* red = reinterpret_cast<float*>(smem + 2 * CI * WPITCH);
    constexpr int RL = 2 * IMG + 2 * CI;
    // the dbias partials: a lane holds the sum over its 8-position groups for channel a*16 + n; add the four lane groups
#pragma unroll
    for (int a = 0; a < MT; ++a) {
      bs2[a] += __shfl_xor(bs2[a], 16, 64); bs2[a] += __shfl_xor(bs2[a], 32, 64);
      bs1[a] += __shfl_xor(bs1[a], 16, 64); bs1[a] += __shfl_xor(bs1[a], 32, 64);
    }
    __syncthreads();
    if (CI == 16) {                                       // the padded tap of the C = 16 images: zeros
      for (int e = tid; e < 2 * CI * CI; e += 256) {
        const int conv = e / (CI * CI), r = e - conv * (CI * CI);
        red[conv * IMG + ((r / CI) * KHP + KHP - 1) * CI + (r % CI)] = 0.f;
      }
    }
#pragma unroll 1
    for (int w = 0; w < 4; ++w) {
      if (wave == w) {
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
          for (int t = 0; t < KR; ++t)
#pragma unroll
            for (int b = 0; b < MT; ++b)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int e = ((a * 16 + g * 4 + r) * KHP + t) * CI + b * 16 + n;
                red[e] = w == 0 ? acc2[a][t][b][r] : red[e] + acc2[a][t][b][r];
                red[IMG + e] = w == 0 ? acc1[a][t][b][r] : red[IMG + e] + acc1[a][t][b][r];
              }
        if (g == 0) {
#pragma unroll
          for (int a = 0; a < MT; ++a) {
            const int e = 2 * IMG + a * 16 + n;
            red[e] = w == 0 ? bs2[a] : red[e] + bs2[a];
            red[e + CI] = w == 0 ? bs1[a] : red[e + CI] + bs1[a];
          }
        }
      }
      __syncthreads();
    }
    float* dst = p.part + (long)blk * p.part_stride;
    for (int e = tid * 4; e < RL; e += 1024) *reinterpret_cast<f32x4*>(dst + e) = *reinterpret_cast<const f32x4*>(red + e);
  }
}

template <int CI, int NK, int NT1, bool WG, int PF>
__global__ __launch_bounds__(256) void resunit_bwd(RBP p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  resunit_bwd_body<CI, NK, NT1, WG, PF>(p, smem, blockIdx.x, gridDim.x);
}

// out[e] += sum over b < nb of part[b * stride + e] for up to 12 (part, out, n) segments in one launch: the images and
// bias gradients of the convolutions of a (grouped) fused backward launch; rows added in block order (fold.hip's scheme)
struct FoldSeg { const float* part; float* out; long n; long stride; int nb; int pad_; };
struct FoldMulti { FoldSeg s[12]; };

__global__ __launch_bounds__(256) void fold_partials_multi(FoldMulti fm) {
  __shared__ float red[16][17];
  const FoldSeg& sg = fm.s[blockIdx.y];
  const int el = threadIdx.x & 15, g = threadIdx.x >> 4;
  const long e = (long)blockIdx.x * 16 + el;
  if ((long)blockIdx.x * 16 >= sg.n) return;
  float s = 0.f;
  if (e < sg.n) {
    int b = g;
    for (; b + 7 * 16 < sg.nb; b += 8 * 16) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = sg.part[(long)(b + u * 16) * sg.stride + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += t[u];
    }
    for (; b < sg.nb; b += 16) s += sg.part[(long)b * sg.stride + e];
  }
  red[g][el] = s;
  __syncthreads();
  if (g == 0 && e < sg.n) {
    float v = red[0][el];
#pragma unroll
    for (int k = 1; k < 16; ++k) v += red[k][el];
    sg.out[e] += v;
  }
}

struct Geo {            // LDS geometry of one problem
  int h1, dyrows, midrows, r_mid, r_xa, r_dm, wave_rows;
  size_t lds;
  int nt1;
};

inline int rup(int v, int m) { return (v + m - 1) / m * m; }

// C, real taps k, dilation -> geometry; false when the unit does not fit
bool geometry(int C, int k, int dil, bool wg, Geo* o) {
  const int khp = C == 16 ? k + 1 : k;
  const int h2 = (k - 1) / 2, h1 = dil * h2;
  const int need = 64 + 2 * h1;
  const int nt1 = (need + 15) / 16;
  if (nt1 > 8) return false;
  // C = 32 with 11 taps: the two gradient images are 352 accumulator registers; the compiler spills ~400 of the 512.
  // That shape runs the data gradients here and its weight gradients as their own launches (dmid is written for them).
  if (wg && C == 32 && k == 11) return false;
  o->nt1 = nt1 <= 5 ? 5 : (nt1 == 6 ? 6 : 8);
  o->h1 = h1;
  o->dyrows = need + khp - 1;
  o->midrows = need;
  const int al = 8;                                  // rows: regions start on 256-byte (C = 16) / 512-byte boundaries
  o->r_mid = rup(o->dyrows, al);
  o->r_xa = o->r_mid + rup(o->midrows, al);
  o->r_dm = o->r_xa + 64;
  int dm = o->nt1 * 16;
  if (64 + (khp - 1) * dil > dm) dm = 64 + (khp - 1) * dil;     // rows a padded tap may touch stay zero
  o->wave_rows = o->r_dm + rup(dm, al);
  const int ktot = khp * C, wpitch = ktot * 2 + 16;
  const size_t wbytes = (size_t)2 * C * wpitch;
  size_t lds = wbytes + (size_t)4 * o->wave_rows * C * 2;
  const size_t red = wbytes + (wg ? ((size_t)2 * C * khp * C + 2 * C) * 4 : 0);
  if (red > lds) lds = red;
  o->lds = lds;
  return lds <= 160 * 1024;
}

// up to three jobs (kernel sizes 3, 7, 11 in that order; a job with no blocks is absent) in one launch.  N3 / N7 / N11:
// stage-A tiles of the three bodies; WG11: the 11-tap job accumulates its weight gradients in the launch (C = 16) or
// writes dmid for separate launches (C = 32)
struct RBPM { RBP job[3]; int blk_end[3]; };

template <int CI, int N3, int N7, int N11, bool WG11>
__global__ __launch_bounds__(256) void resunit_bwd_multi(RBPM pm) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  constexpr int NK3 = CI == 16 ? 2 : 3, NK7 = CI == 16 ? 4 : 7, NK11 = CI == 16 ? 6 : 11;
  constexpr int PF = CI == 16 ? 1 : 0;
  const int b = blockIdx.x;
  if (b < pm.blk_end[0]) resunit_bwd_body<CI, NK3, N3, true, PF>(pm.job[0], smem, b, pm.blk_end[0]);
  else if (b < pm.blk_end[1]) resunit_bwd_body<CI, NK7, N7, true, PF>(pm.job[1], smem, b - pm.blk_end[0], pm.blk_end[1] - pm.blk_end[0]);
  else resunit_bwd_body<CI, NK11, N11, WG11, PF>(pm.job[2], smem, b - pm.blk_end[1], pm.blk_end[2] - pm.blk_end[1]);
}

template <typename K>
int set_lds_once(K kernel, bool* done) {
  if (!*done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return EVT_ELAUNCH;
    *done = true;
  }
  return EVT_OK;
}

template <int CI, int N3, int N7, int N11>
int launch_multi(const RBPM& pm, size_t lds, int njobs, hipStream_t st) {
  constexpr bool WG11 = CI == 16;
  static bool attr = false;
  if (set_lds_once(&resunit_bwd_multi<CI, N3, N7, N11, WG11>, &attr)) return EVT_ELAUNCH;
  evt_set_last_tag("resunit_bwd_multi<bf16, %d, x%d, nt %d-%d-%d>", CI, njobs, N3, N7, N11);
  hipLaunchKernelGGL((resunit_bwd_multi<CI, N3, N7, N11, WG11>), dim3(pm.blk_end[2]), dim3(256), lds, st, pm);
  return evt_check_launch();
}

template <int CI>
int launch_multi_nt(const RBPM& pm, const int (&nt)[3], size_t lds, int njobs, hipStream_t st) {
  // the three instantiations cover every (kernel size, dilation <= 5) combination: a job may run with more stage-A tiles
  // than it needs (the surplus rows are computed and dropped)
  if (nt[0] <= 5 && nt[1] <= 5 && nt[2] <= 5) return launch_multi<CI, 5, 5, 5>(pm, lds, njobs, st);
  if (nt[0] <= 5 && nt[1] <= 6 && nt[2] <= 6) return launch_multi<CI, 5, 6, 6>(pm, lds, njobs, st);
  if (nt[0] <= 5 && nt[1] <= 6 && nt[2] <= 8) return launch_multi<CI, 5, 6, 8>(pm, lds, njobs, st);
  return EVT_ENOTSUP;
}

}  // namespace

extern "C" {

int32_t evt_resunit_bwd_supported(const evt_resunit_params* a, int32_t with_weight_grads) {
  static const bool off = getenv("EVT_NO_RESUNIT_BWD") != nullptr;   // A/B switch for measurements
  if (off || !evt_resunit_supported(a)) return 0;
  Geo geo;
  return geometry(a->C, a->k, a->dil, with_weight_grads != 0, &geo) ? 1 : 0;
}

int64_t evt_resunit_bwd_ws_floats(const evt_resunit_params* a) {
  if (!evt_resunit_bwd_supported(a, 0)) return 0;
  const int khp = a->C == 16 ? a->k + 1 : a->k;
  return (int64_t)1024 * (2L * a->C * khp * a->C + 2 * a->C);
}

// One to three steps (kernel sizes 3 / 7 / 11, one job each, same C) in one launch + one fold launch.  A job whose
// weight gradients the kernel cannot hold (C = 32, 11 taps) must come with dmid != NULL and dw1 == dw2 == NULL: the
// caller runs evt_conv1d_bwd_weight for it.
int evt_resunit_bwd_multi(const evt_resunit_bwd_job* jobs, int32_t njobs, float* ws, int64_t ws_floats, void* stream) {
  if (!jobs || njobs < 1 || njobs > 3) return EVT_EINVAL;
  static const bool multi_off = getenv("EVT_NO_RESUNIT_MULTI") != nullptr;   // A/B switch for measurements
  if (multi_off && njobs > 1) return EVT_ENOTSUP;
  RBPM pm{};
  const int C = jobs[0].p.C;
  size_t lds = 0;
  double cost[3] = {0, 0, 0};
  long row[3] = {0, 0, 0}, img[3] = {0, 0, 0};
  int nt[3] = {5, 5, 5};
  const evt_resunit_bwd_job* js[3] = {nullptr, nullptr, nullptr};
  for (int j = 0; j < njobs; ++j) {
    const evt_resunit_bwd_job& jb = jobs[j];
    if (jb.p.C != C) return EVT_ENOTSUP;
    const bool wg = jb.dw1 != nullptr || jb.dw2 != nullptr;
    if (!evt_resunit_bwd_supported(&jb.p, wg)) return EVT_ENOTSUP;
    if (!jb.dy || !jb.xa || !jb.mid_a || !jb.w1_alt || !jb.w2_alt || !jb.dx) return EVT_EINVAL;
    if (wg && (!jb.dw1 || !jb.dw2 || !ws)) return EVT_EINVAL;
    const int slot = jb.p.k == 3 ? 0 : (jb.p.k == 7 ? 1 : 2);
    if (js[slot]) return EVT_ENOTSUP;                     // one job per kernel size
    // what the instantiation of this slot accumulates: everything but the 11-tap job at C = 32
    const bool slot_wg = !(slot == 2 && C == 32);
    if (wg != slot_wg) return EVT_ENOTSUP;
    js[slot] = &jb;
    Geo geo;
    geometry(C, jb.p.k, jb.p.dil, wg, &geo);
    RBP& p = pm.job[slot];
    p.dy = (const h16_t*)jb.dy; p.xa = (const h16_t*)jb.xa; p.mid = (const h16_t*)jb.mid_a;
    p.w1 = (const h16_t*)jb.w1_alt; p.w2 = (const h16_t*)jb.w2_alt; p.dx = (h16_t*)jb.dx; p.dmid = (h16_t*)jb.dmid;
    p.nseq = jb.p.nseq; p.L = jb.p.L; p.dil = jb.p.dil; p.slope = jb.p.slope; p.dy_scale = jb.dy_scale;
    p.ups = (jb.p.L + 63) / 64;
    p.total = (long)jb.p.nseq * p.ups;
    p.h1 = geo.h1; p.dyrows = geo.dyrows; p.midrows = geo.midrows;
    p.r_mid = geo.r_mid; p.r_xa = geo.r_xa; p.r_dm = geo.r_dm; p.wave_rows = geo.wave_rows;
    nt[slot] = geo.nt1;
    if (geo.lds > lds) lds = geo.lds;
    const int khp = C == 16 ? jb.p.k + 1 : jb.p.k;
    img[slot] = (long)C * khp * C;
    row[slot] = wg ? 2 * img[slot] + 2 * C : 0;
    // measured single launches (us, B = 16): C = 16: 22 / 32 / 50, C = 32: 35 / 56 / 72
    cost[slot] = (double)p.total * (1.0 + 0.35 * jb.p.k);
  }
  // the instantiation may give a job more stage-A tiles than its geometry asked for: their rows are computed from
  // whatever lies behind the staged ones and forced to zero, but they ARE written -- the DM region must hold them
  const bool all5 = nt[0] <= 5 && nt[1] <= 5 && nt[2] <= 5, all6 = nt[0] <= 5 && nt[1] <= 6 && nt[2] <= 6;
  const int inst_nt[3] = {5, all5 ? 5 : 6, all5 ? 5 : (all6 ? 6 : 8)};   // = launch_multi_nt's choice
  for (int s = 0; s < 3; ++s) {
    if (!js[s]) continue;
    RBP& p = pm.job[s];
    const int khp = C == 16 ? js[s]->p.k + 1 : js[s]->p.k;
    if (rup(inst_nt[s] * 16, 8) > p.wave_rows - p.r_dm) {
      p.wave_rows = p.r_dm + rup(inst_nt[s] * 16, 8);
      const size_t wbytes = (size_t)2 * C * (khp * C * 2 + 16);
      const size_t l = wbytes + (size_t)4 * p.wave_rows * C * 2;
      if (l > lds) lds = l;
    }
  }
  if (lds > 160 * 1024) return EVT_ENOTSUP;
  const int per_cu = (int)((160 * 1024) / lds) < 2 ? (int)((160 * 1024) / lds) : 2;
  static const long cap_env = getenv("EVT_RESUNIT_BWD_BLOCKS") ? atol(getenv("EVT_RESUNIT_BWD_BLOCKS")) : 0;
  long cap = cap_env > 0 ? cap_env : 256L * per_cu;
  const double tot = cost[0] + cost[1] + cost[2];
  // the partial rows must fit the scratch
  for (;;) {
    double fl = 0;
    for (int s = 0; s < 3; ++s) fl += cap * cost[s] / tot * row[s] + row[s];
    if (fl <= (double)ws_floats || cap <= 8) break;
    cap = cap * 3 / 4;
  }
  int end = 0;
  long off[3] = {0, 0, 0}, woff = 0;
  int nb[3] = {0, 0, 0};
  for (int s = 0; s < 3; ++s) {
    if (js[s]) {
      long b = (long)(cap * cost[s] / tot + 0.5);
      const long need = (pm.job[s].total + 3) / 4;
      if (b > need) b = need;
      if (b < 1) b = 1;
      nb[s] = (int)b;
      end += (int)b;
      off[s] = woff;
      if (row[s]) {
        if (woff + b * row[s] > ws_floats) return EVT_EINVAL;
        pm.job[s].part = ws + woff;
        pm.job[s].part_stride = row[s];
        woff += b * row[s];
      }
    }
    pm.blk_end[s] = end;
  }
  hipStream_t st = (hipStream_t)stream;
  int rc = C == 16 ? launch_multi_nt<16>(pm, nt, lds, njobs, st) : launch_multi_nt<32>(pm, nt, lds, njobs, st);
  if (rc) return rc;
  FoldMulti fm{};
  int ns = 0;
  long maxn = 0;
  for (int s = 0; s < 3; ++s) {
    if (!js[s] || !row[s]) continue;
    float* base = ws + off[s];
    fm.s[ns++] = FoldSeg{base, js[s]->dw2, img[s], row[s], nb[s], 0};
    fm.s[ns++] = FoldSeg{base + img[s], js[s]->dw1, img[s], row[s], nb[s], 0};
    if (js[s]->db2) fm.s[ns++] = FoldSeg{base + 2 * img[s], js[s]->db2, C, row[s], nb[s], 0};
    if (js[s]->db1) fm.s[ns++] = FoldSeg{base + 2 * img[s] + C, js[s]->db1, C, row[s], nb[s], 0};
    if (img[s] > maxn) maxn = img[s];
  }
  if (!ns) return EVT_OK;
  hipLaunchKernelGGL(fold_partials_multi, dim3((unsigned)((maxn + 15) / 16), ns), dim3(256), 0, st, fm);
  return evt_check_launch();
}

int evt_resunit_bwd(const evt_resunit_params* a, const void* dy, float dy_scale, const void* xa, const void* mid_a,
                    const void* w1_alt, const void* w2_alt, void* dx, void* dmid, float* dw1, float* dw2, float* db1,
                    float* db2, float* ws, int64_t ws_floats, void* stream) {
  if (!a) return EVT_EINVAL;
  evt_resunit_bwd_job jb{};
  jb.p = *a; jb.dy = dy; jb.dy_scale = dy_scale; jb.xa = xa; jb.mid_a = mid_a; jb.w1_alt = w1_alt; jb.w2_alt = w2_alt;
  jb.dx = dx; jb.dmid = dmid; jb.dw1 = dw1; jb.dw2 = dw2; jb.db1 = db1; jb.db2 = db2;
  return evt_resunit_bwd_multi(&jb, 1, ws, ws_floats, stream);
}

}  // extern "C"
