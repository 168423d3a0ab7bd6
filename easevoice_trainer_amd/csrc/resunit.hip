// resunit_fwd: one HiFi-GAN ResBlock1 step  y = x + c2(lrelu(c1(lrelu(x))))  (src/easevoice/module/modules.py:299-308 of
// the reference; c1 dilated, c2 dilation 1, both C -> C, "same" padding) as ONE launch for the narrow vocoder stages
// (C = 16 at 20480 samples, C = 32 at 10240), gfx950 bf16 -- and, grouped, the same step of the THREE ResBlocks of a
// stage (k = 3 / 7 / 11, models.py:457-466) as one launch.
//
// These stages move 21 MB per convolution for 0.25-3.7 GMAC: the HBM-bound end of the vocoder.  Unfused, a step is a
// leaky-relu launch + two conv launches (7 tensor passes); here x is read once and y, lrelu(x) and lrelu(c1(..)) (the two
// tensors the backward launch wants) are written once (4 passes), and the activated intermediate never leaves LDS
// before c2 consumes it:
//   * both weight matrices sit in LDS for the life of the block ([co][tap][ci] rows, odd 16-byte pitch);
//   * waves are independent and persistent over 64-position units: a wave stages its own input rows (+ the halo of BOTH
//     convolutions, activated on the way in) in a private LDS region (layout: resunit_common.h), computes c1 on 80
//     positions (64 + the c2 halo) into a private LDS tile (bias, leaky-relu, zero outside the sequence = c2's zero
//     padding), then c2 on the 64 positions; the next unit's rows are prefetched into registers under the MFMAs;
//   * both convolutions are the software-pipelined conv_stage of resunit_common.h (fragments of K step s + 1 requested
//     before the MFMAs of step s); the residual rows are requested before c2's MFMAs;
//   * epilogues: 8-byte stores tiling whole position rows.
// A grouped launch hands each of its (up to three) jobs a range of blocks; a block runs the body of its job's kernel
// size.  Three 20-30 us launches whose length is mostly fill and drain become one.
#include "resunit_common.h"
#include "../../include/evt.h"
#include <cstdlib>

namespace {

using namespace evt_ru;

struct RUP {
  const h16_t* x; const h16_t* w1; const h16_t* w2; const float* b1; const float* b2;
  h16_t* xa; h16_t* mid; h16_t* y;
  int nseq, L, k, dil;
  float slope;
  int xrows;            // staged input rows per unit
  int r_ms;             // first row of the intermediate tile inside a wave's area
  int wave_rows;        // rows per wave
  int ups;              // units per sequence
  long total;           // units
};

// CI: channels (16 or 32); NK: K steps of 32 per convolution = padded taps * CI / 32
template <int CI, int NK>
__device__ __forceinline__ void resunit_fwd_body(const RUP& p, unsigned char* smem, const int blk, const int nblk) {
  constexpr int MT = CI / 16;
  constexpr int PITCH = CI * 2;
  constexpr int LOGP = CI == 16 ? 1 : 2;
  constexpr int PPR = 1 << LOGP;
  constexpr int KTOT = NK * 32;                           // elements per weight row [tap][ci], zero padded (KTOT / CI taps)
  constexpr int KHP = KTOT / CI;
  constexpr int KR = CI == 16 ? KHP - 1 : KHP;
  constexpr int H2 = (KR - 1) / 2;
  constexpr int WPITCH = KTOT * 2 + 16;                   // weight row pitch: odd number of 16-byte slots
  constexpr int NT1 = 5, NT2 = 4;                         // position tiles of 16: c1 on 80, c2 on 64
  constexpr int XPT = ((80 + (KHP - 1) * 5) * PPR + 63) / 64;   // prefetch registers for the largest dilation (5)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  unsigned char* wl1 = smem;
  unsigned char* wl2 = smem + CI * WPITCH;
  unsigned char* xs = smem + 2 * CI * WPITCH + wave * p.wave_rows * PITCH;
  unsigned char* ms = xs + p.r_ms * PITCH;

  load_weights<CI, KTOT>(wl1, wl2, p.w1, p.w2);
  float b1[MT][4], b2[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) { b1[i][r] = p.b1 ? p.b1[i * 16 + g * 4 + r] : 0.f; b2[i][r] = p.b2 ? p.b2[i * 16 + g * 4 + r] : 0.f; }
  __syncthreads();

  const int h1 = p.dil * H2;
  const long wave_id = (long)blk * 4 + wave, nwaves = (long)nblk * 4;
  const int npieces = p.xrows * PPR;
  uint4 xr[XPT];
  auto load_unit = [&](long u) {
    const int seq = (int)(u / p.ups);
    const int row0 = (int)(u - (long)seq * p.ups) * 64 - H2 - h1;
    const h16_t* xg = p.x + (long)seq * p.L * CI;
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int idx = lane + i * 64;
      const int r = idx >> LOGP, part = idx & (PPR - 1);
      const int in_row = row0 + r;
      const bool ok = idx < npieces && in_row >= 0 && in_row < p.L;
      xr[i] = ok ? *reinterpret_cast<const uint4*>(xg + (long)in_row * CI + part * 8) : make_uint4(0, 0, 0, 0);
    }
  };
  if (wave_id < p.total) load_unit(wave_id);
  for (long u = wave_id; u < p.total; u += nwaves) {
    const int seq = (int)(u / p.ups);
    const int q0 = (int)(u - (long)seq * p.ups) * 64;
    const long sbase = (long)seq * p.L * CI;
    // activate and publish this unit's rows (previous unit's fragment reads are complete: in-order DS); the unit's own
    // 64 rows of lrelu(x) also go to global for the backward launch
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int idx = lane + i * 64;
      if (idx < npieces) {
        const int r = idx >> LOGP, part = idx & (PPR - 1);
        const uint4 a = lrelu8(xr[i], p.slope);
        *reinterpret_cast<uint4*>(xs + piece_off<CI>(r, part)) = a;
        const int o = r - h1 - H2;                    // position q0 + o
        if (p.xa && o >= 0 && o < 64 && q0 + o < p.L)
          *reinterpret_cast<uint4*>(p.xa + sbase + (long)(q0 + o) * CI + part * 8) = a;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (u + nwaves < p.total) load_unit(u + nwaves);       // flies under the MFMAs below

    // ---- c1 on the 80 positions m0 + [0, 80), m0 = q0 - H2: region row of (position row, tap) = row + tap * dil ----
    {
      f32x4 acc[MT][NT1];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT1; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      conv_stage<CI, NK, MT, NT1>(acc, wl1, xs, p.dil, n, g);
      // bias, leaky-relu, zero outside the sequence (c2 pads its INPUT with zeros); lane: channels i*16+g*4.., position j*16+n
#pragma unroll
      for (int j = 0; j < NT1; ++j) {
        const int pos = j * 16 + n;
        const int m = q0 - H2 + pos;
        const bool inside = m >= 0 && m < p.L;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          h16_t o4[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = acc[i][j][r] + b1[i][r];
            v = v > 0.f ? v : v * p.slope;
            o4[r] = f2h(inside ? v : 0.f);
          }
          *reinterpret_cast<uint2*>(ms + chan_off<CI>(pos, i * 16 + g * 4)) = *reinterpret_cast<uint2*>(o4);
          if (p.mid && inside && pos >= H2 && pos < H2 + 64)
            *reinterpret_cast<uint2*>(p.mid + sbase + (long)m * CI + i * 16 + g * 4) = *reinterpret_cast<uint2*>(o4);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();

    // ---- c2 (dilation 1) on the 64 positions q0 + [0, 64): tile row of (position row, tap) = row + tap ----
    {
      f32x4 acc[MT][NT2];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      // the residual rows (L2-hot: this wave loaded them a unit ago) travel under the MFMAs
      uint2 rv[NT2][MT];
#pragma unroll
      for (int j = 0; j < NT2; ++j) {
        const int q = q0 + j * 16 + n;
#pragma unroll
        for (int i = 0; i < MT; ++i)
          rv[j][i] = q < p.L ? *reinterpret_cast<const uint2*>(p.x + sbase + (long)q * CI + i * 16 + g * 4) : make_uint2(0, 0);
      }
      conv_stage<CI, NK, MT, NT2>(acc, wl2, ms, 1, n, g);
#pragma unroll
      for (int j = 0; j < NT2; ++j) {
        const int q = q0 + j * 16 + n;
        if (q >= p.L) continue;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const h16_t* pr = reinterpret_cast<const h16_t*>(&rv[j][i]);
          h16_t o4[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) o4[r] = f2h(acc[i][j][r] + b2[i][r] + h2f(pr[r]));
          *reinterpret_cast<uint2*>(p.y + sbase + (long)q * CI + i * 16 + g * 4) = *reinterpret_cast<uint2*>(o4);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
}

template <int CI, int NK>
__global__ __launch_bounds__(256) void resunit_fwd(RUP p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  resunit_fwd_body<CI, NK>(p, smem, blockIdx.x, gridDim.x);
}

// up to three jobs (kernel sizes 3, 7, 11 in that order; a job with no blocks is absent) in one launch
struct RUPM { RUP job[3]; int blk_end[3]; };

template <int CI>
__global__ __launch_bounds__(256) void resunit_fwd_multi(RUPM pm) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  constexpr int NK3 = CI == 16 ? 2 : 3, NK7 = CI == 16 ? 4 : 7, NK11 = CI == 16 ? 6 : 11;
  const int b = blockIdx.x;
  if (b < pm.blk_end[0]) resunit_fwd_body<CI, NK3>(pm.job[0], smem, b, pm.blk_end[0]);
  else if (b < pm.blk_end[1]) resunit_fwd_body<CI, NK7>(pm.job[1], smem, b - pm.blk_end[0], pm.blk_end[1] - pm.blk_end[0]);
  else resunit_fwd_body<CI, NK11>(pm.job[2], smem, b - pm.blk_end[1], pm.blk_end[2] - pm.blk_end[1]);
}

inline int rup8(int v) { return (v + 7) / 8 * 8; }

// fills the geometry of one job; returns its LDS bytes (0: does not fit)
size_t fwd_geometry(RUP& p, int C) {
  const int khp = C == 16 ? p.k + 1 : p.k;
  p.xrows = 80 + (khp - 1) * p.dil;
  p.r_ms = rup8(p.xrows);
  p.wave_rows = p.r_ms + 80;
  p.ups = (p.L + 63) / 64;
  p.total = (long)p.nseq * p.ups;
  const int wpitch = khp * C * 2 + 16;
  const size_t lds = (size_t)2 * C * wpitch + (size_t)4 * p.wave_rows * C * 2;
  return lds <= 160 * 1024 ? lds : 0;
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

template <int CI, int NK>
int launch(const RUP& p, size_t lds, hipStream_t st) {
  const int per_cu = (int)((160 * 1024) / lds) < 4 ? (int)((160 * 1024) / lds) : 4;
  long blocks = (p.total + 3) / 4;
  static const long cap_env = getenv("EVT_RESUNIT_BLOCKS") ? atol(getenv("EVT_RESUNIT_BLOCKS")) : 0;   // tuning knob
  const long cap = cap_env > 0 ? cap_env : 256L * per_cu;
  if (blocks > cap) blocks = cap;
  static bool attr = false;
  if (set_lds_once(&resunit_fwd<CI, NK>, &attr)) return EVT_ELAUNCH;
  evt_set_last_tag("resunit_fwd<bf16, %d, k%d>", CI, p.k);
  hipLaunchKernelGGL((resunit_fwd<CI, NK>), dim3((int)blocks), dim3(256), lds, st, p);
  return evt_check_launch();
}

bool job_ok(const evt_resunit_params* a) {
  if (!a || a->dtype != EVT_DT_HALF) return false;
  if (a->C != 16 && a->C != 32) return false;
  if (a->k != 3 && a->k != 7 && a->k != 11) return false;
  if (a->dil < 1 || a->dil > 5 || a->nseq <= 0 || a->L < 64) return false;
  return true;
}

}  // namespace

extern "C" {

int32_t evt_resunit_supported(const evt_resunit_params* a) {
  static const bool off = getenv("EVT_NO_RESUNIT") != nullptr;   // A/B switch for measurements
  return (!off && job_ok(a)) ? 1 : 0;
}

int evt_resunit_fwd(const evt_resunit_params* a, const void* x, const void* w1_reg, const void* w2_reg, const float* b1,
                    const float* b2, void* xa, void* mid_a, void* y, void* stream) {
  if (!evt_resunit_supported(a)) return EVT_ENOTSUP;
  if (!x || !w1_reg || !w2_reg || !y) return EVT_EINVAL;
  RUP p{};
  p.x = (const h16_t*)x; p.w1 = (const h16_t*)w1_reg; p.w2 = (const h16_t*)w2_reg; p.b1 = b1; p.b2 = b2;
  p.xa = (h16_t*)xa; p.mid = (h16_t*)mid_a; p.y = (h16_t*)y;
  p.nseq = a->nseq; p.L = a->L; p.k = a->k; p.dil = a->dil; p.slope = a->slope;
  const size_t lds = fwd_geometry(p, a->C);
  if (!lds) return EVT_ENOTSUP;
  hipStream_t st = (hipStream_t)stream;
  if (a->C == 16) {
    if (a->k == 3) return launch<16, 2>(p, lds, st);
    if (a->k == 7) return launch<16, 4>(p, lds, st);
    return launch<16, 6>(p, lds, st);
  }
  if (a->k == 3) return launch<32, 3>(p, lds, st);
  if (a->k == 7) return launch<32, 7>(p, lds, st);
  return launch<32, 11>(p, lds, st);
}

int evt_resunit_fwd_multi(const evt_resunit_fwd_job* jobs, int32_t njobs, void* stream) {
  if (!jobs || njobs < 1 || njobs > 3) return EVT_EINVAL;
  static const bool off = getenv("EVT_NO_RESUNIT_MULTI") != nullptr;   // A/B switch for measurements
  if (off) return EVT_ENOTSUP;
  RUPM pm{};
  size_t lds = 0;
  double cost[3] = {0, 0, 0};
  const int C = jobs[0].p.C;
  for (int j = 0; j < njobs; ++j) {
    const evt_resunit_fwd_job& jb = jobs[j];
    if (!evt_resunit_supported(&jb.p) || jb.p.C != C) return EVT_ENOTSUP;
    if (!jb.x || !jb.w1_reg || !jb.w2_reg || !jb.y) return EVT_EINVAL;
    const int slot = jb.p.k == 3 ? 0 : (jb.p.k == 7 ? 1 : 2);
    if (cost[slot] != 0) return EVT_ENOTSUP;              // one job per kernel size
    RUP& p = pm.job[slot];
    p.x = (const h16_t*)jb.x; p.w1 = (const h16_t*)jb.w1_reg; p.w2 = (const h16_t*)jb.w2_reg; p.b1 = jb.b1; p.b2 = jb.b2;
    p.xa = (h16_t*)jb.xa; p.mid = (h16_t*)jb.mid_a; p.y = (h16_t*)jb.y;
    p.nseq = jb.p.nseq; p.L = jb.p.L; p.k = jb.p.k; p.dil = jb.p.dil; p.slope = jb.p.slope;
    const size_t l = fwd_geometry(p, C);
    if (!l) return EVT_ENOTSUP;
    if (l > lds) lds = l;
    cost[slot] = (double)p.total * (1.0 + 0.12 * jb.p.k);  // measured single launches: 20 / 21 / 21.5 us (C = 16), 21 / 25 / 29 (C = 32)
  }
  const int per_cu = (int)((160 * 1024) / lds) < 4 ? (int)((160 * 1024) / lds) : 4;
  static const long cap_env = getenv("EVT_RESUNIT_BLOCKS") ? atol(getenv("EVT_RESUNIT_BLOCKS")) : 0;
  const long cap = cap_env > 0 ? cap_env : 256L * per_cu;
  const double tot = cost[0] + cost[1] + cost[2];
  int end = 0;
  for (int s = 0; s < 3; ++s) {
    if (cost[s] > 0) {
      long nb = (long)(cap * cost[s] / tot + 0.5);
      const long need = (pm.job[s].total + 3) / 4;
      if (nb > need) nb = need;
      if (nb < 1) nb = 1;
      end += (int)nb;
    }
    pm.blk_end[s] = end;
  }
  hipStream_t st = (hipStream_t)stream;
  if (C == 16) {
    static bool attr = false;
    if (set_lds_once(&resunit_fwd_multi<16>, &attr)) return EVT_ELAUNCH;
    evt_set_last_tag("resunit_fwd_multi<bf16, 16, x%d>", njobs);
    hipLaunchKernelGGL((resunit_fwd_multi<16>), dim3(end), dim3(256), lds, st, pm);
  } else {
    static bool attr = false;
    if (set_lds_once(&resunit_fwd_multi<32>, &attr)) return EVT_ELAUNCH;
    evt_set_last_tag("resunit_fwd_multi<bf16, 32, x%d>", njobs);
    hipLaunchKernelGGL((resunit_fwd_multi<32>), dim3(end), dim3(256), lds, st, pm);
  }
  return evt_check_launch();
}

}  // extern "C"
