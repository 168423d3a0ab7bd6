// resunit_fwd: one HiFi-GAN ResBlock1 step  y = x + c2(lrelu(c1(lrelu(x))))  (src/easevoice/module/modules.py:299-308 of
// the reference; c1 dilated, c2 dilation 1, both C -> C, "same" padding) as ONE launch for the narrow vocoder stages
// (C = 16 at 20480 samples, C = 32 at 10240), gfx950 bf16.
//
// These stages move 21 MB per convolution for 0.25-3.7 GMAC: the HBM-bound end of the vocoder.  Unfused, a step is a
// leaky-relu launch + two conv launches (7 tensor passes); here x is read once and y, lrelu(x) and lrelu(c1(..)) (the two
// tensors the backward launches want) are written once (4 passes), and the activated intermediate never leaves LDS
// before c2 consumes it:
//   * both weight matrices sit in LDS for the life of the block ([co][tap][ci] rows, odd 16-byte pitch), A fragments are
//     re-read per K step (one 16-byte ds_read per 16 output channels, shared by all position tiles of the step);
//   * waves are independent and persistent over 64-position units, like conv_narrow: a wave stages its own input rows
//     (+ the halo of BOTH convolutions, activated on the way in) in a private LDS region, computes c1 on 80 positions
//     (64 + the c2 halo) into a private LDS tile (bias, leaky-relu, zero outside the sequence = c2's zero padding),
//     then c2 on the 64 positions; the next unit's rows are prefetched into registers under the MFMAs;
//   * epilogues: 8-byte stores tiling whole position rows; the residual is re-read from global (L2-hot).
#include "evt_common.h"
#include "../../include/evt.h"
#include <cstdlib>

namespace {

struct RUP {
  const bf16_t* x; const bf16_t* w1; const bf16_t* w2; const float* b1; const float* b2;
  bf16_t* xa; bf16_t* mid; bf16_t* y;
  int nseq, L, k, dil;
  float slope;
  int xrows;            // staged input rows per unit
  int ups;              // units per sequence
  long total;           // units
};

__device__ __forceinline__ uint32_t lrelu2(uint32_t d, float slope) {      // two packed bf16
  float a = __uint_as_float(d << 16), b = __uint_as_float(d & 0xFFFF0000u);
  a = a > 0.f ? a : a * slope;
  b = b > 0.f ? b : b * slope;
  return (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16);
}
__device__ __forceinline__ uint4 lrelu8(uint4 v, float slope) {
  return make_uint4(lrelu2(v.x, slope), lrelu2(v.y, slope), lrelu2(v.z, slope), lrelu2(v.w, slope));
}

// CI: channels (16 or 32); NK: K steps of 32 per convolution = padded taps * CI / 32
template <int CI, int NK>
__global__ __launch_bounds__(256) void resunit_fwd(RUP p) {
  constexpr int MT = CI / 16;
  constexpr int XROW = CI == 16 ? 32 : 96;               // activation row pitch (bytes): conflict-free 16-byte reads
  constexpr int PPR = CI * 2 / 16;                        // 16-byte pieces per row
  constexpr int KTOT = NK * 32;                           // elements per weight row [tap][ci], zero padded (KTOT / CI taps)
  constexpr int WPITCH = KTOT * 2 + 16;                   // weight row pitch: odd number of 16-byte slots
  constexpr int NT1 = 5, NT2 = 4;                         // position tiles of 16: c1 on 80, c2 on 64
  constexpr int XPT = CI == 16 ? 5 : 9;                   // prefetch registers (xrows * PPR <= 64 * XPT)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  unsigned char* wl1 = smem;
  unsigned char* wl2 = smem + CI * WPITCH;
  unsigned char* xs = smem + 2 * CI * WPITCH + wave * (p.xrows + 80) * XROW;
  unsigned char* ms = xs + p.xrows * XROW;

  // weights -> LDS (prepared images [co][KHP][CI], K index contiguous)
  for (int idx = tid; idx < CI * (KTOT / 8); idx += 256) {
    const int co = idx / (KTOT / 8), part = idx - co * (KTOT / 8);
    *reinterpret_cast<uint4*>(wl1 + co * WPITCH + part * 16) = *reinterpret_cast<const uint4*>(p.w1 + (long)co * KTOT + part * 8);
    *reinterpret_cast<uint4*>(wl2 + co * WPITCH + part * 16) = *reinterpret_cast<const uint4*>(p.w2 + (long)co * KTOT + part * 8);
  }
  float b1[MT][4], b2[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) { b1[i][r] = p.b1 ? p.b1[i * 16 + g * 4 + r] : 0.f; b2[i][r] = p.b2 ? p.b2[i * 16 + g * 4 + r] : 0.f; }
  __syncthreads();

  const int h2 = (p.k - 1) / 2, h1 = p.dil * (p.k - 1) / 2;
  const long wave_id = (long)blockIdx.x * 4 + wave, nwaves = (long)gridDim.x * 4;
  const int npieces = p.xrows * PPR;
  uint4 xr[XPT];
  auto load_unit = [&](long u) {
    const int seq = (int)(u / p.ups);
    const int row0 = (int)(u - (long)seq * p.ups) * 64 - h2 - h1;
    const bf16_t* xg = p.x + (long)seq * p.L * CI;
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int idx = lane + i * 64;
      const int r = idx / PPR, part = idx - r * PPR;
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
    // 64 rows of lrelu(x) also go to global for the backward launches
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int idx = lane + i * 64;
      if (idx < npieces) {
        const int r = idx / PPR, part = idx - r * PPR;
        const uint4 a = lrelu8(xr[i], p.slope);
        *reinterpret_cast<uint4*>(xs + r * XROW + part * 16) = a;
        const int o = r - h1 - h2;                    // position q0 + o
        if (p.xa && o >= 0 && o < 64 && q0 + o < p.L)
          *reinterpret_cast<uint4*>(p.xa + sbase + (long)(q0 + o) * CI + part * 8) = a;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (u + nwaves < p.total) load_unit(u + nwaves);       // flies under the MFMAs below

    // ---- c1 on the 80 positions m0 + [0, 80), m0 = q0 - h2: region row of (position row, tap) = row + tap * dil ----
    {
      f32x4 acc[MT][NT1];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT1; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NK; ++ks) {
        const int kk = ks * 32 + g * 8;
        const int tap = kk / CI, ci = kk - tap * CI;
        bf16x8 af[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const bf16x8*>(wl1 + (i * 16 + n) * WPITCH + kk * 2);
        const unsigned char* base = xs + (n + tap * p.dil) * XROW + ci * 2;
#pragma unroll
        for (int j = 0; j < NT1; ++j) {
          const bf16x8 b = *reinterpret_cast<const bf16x8*>(base + j * 16 * XROW);
#pragma unroll
          for (int i = 0; i < MT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], b, acc[i][j], 0, 0, 0);
        }
      }
      // bias, leaky-relu, zero outside the sequence (c2 pads its INPUT with zeros); lane: channels i*16+g*4.., position j*16+n
#pragma unroll
      for (int j = 0; j < NT1; ++j) {
        const int pos = j * 16 + n;
        const int m = q0 - h2 + pos;
        const bool inside = m >= 0 && m < p.L;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          bf16_t o4[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = acc[i][j][r] + b1[i][r];
            v = v > 0.f ? v : v * p.slope;
            o4[r] = f2bf(inside ? v : 0.f);
          }
          *reinterpret_cast<uint2*>(ms + pos * XROW + (i * 16 + g * 4) * 2) = *reinterpret_cast<uint2*>(o4);
          if (p.mid && inside && pos >= h2 && pos < h2 + 64)
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
#pragma unroll
      for (int ks = 0; ks < NK; ++ks) {
        const int kk = ks * 32 + g * 8;
        const int tap = kk / CI, ci = kk - tap * CI;
        bf16x8 af[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const bf16x8*>(wl2 + (i * 16 + n) * WPITCH + kk * 2);
        const unsigned char* base = ms + (n + tap) * XROW + ci * 2;
#pragma unroll
        for (int j = 0; j < NT2; ++j) {
          const bf16x8 b = *reinterpret_cast<const bf16x8*>(base + j * 16 * XROW);
#pragma unroll
          for (int i = 0; i < MT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], b, acc[i][j], 0, 0, 0);
        }
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < NT2; ++j) {
        const int q = q0 + j * 16 + n;
        if (q >= p.L) continue;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const long off = sbase + (long)q * CI + i * 16 + g * 4;
          const uint2 rv = *reinterpret_cast<const uint2*>(p.x + off);
          const bf16_t* pr = reinterpret_cast<const bf16_t*>(&rv);
          bf16_t o4[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) o4[r] = f2bf(acc[i][j][r] + b2[i][r] + bf2f(pr[r]));
          *reinterpret_cast<uint2*>(p.y + off) = *reinterpret_cast<uint2*>(o4);
        }
      }
    }
  }
}

template <int CI, int NK>
int launch(const RUP& p0, hipStream_t st) {
  RUP p = p0;
  constexpr int XROW = CI == 16 ? 32 : 96;
  constexpr int KTOT = NK * 32, KHP = KTOT / CI, WPITCH = KTOT * 2 + 16;
  constexpr int XPT = CI == 16 ? 5 : 9;
  p.xrows = 79 + (KHP - 1) * p.dil + 1;
  if (p.xrows * (CI * 2 / 16) > 64 * XPT) return EVT_ENOTSUP;
  p.ups = (p.L + 63) / 64;
  p.total = (long)p.nseq * p.ups;
  const size_t lds = (size_t)2 * CI * WPITCH + (size_t)4 * (p.xrows + 80) * XROW;
  if (lds > 160 * 1024) return EVT_ENOTSUP;
  const int per_cu = (int)((160 * 1024) / lds) < 4 ? (int)((160 * 1024) / lds) : 4;
  long blocks = (p.total + 3) / 4;
  static const long cap_env = getenv("EVT_RESUNIT_BLOCKS") ? atol(getenv("EVT_RESUNIT_BLOCKS")) : 0;   // tuning knob
  const long cap = cap_env > 0 ? cap_env : 256L * per_cu;
  if (blocks > cap) blocks = cap;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&resunit_fwd<CI, NK>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return EVT_ELAUNCH;
    attr = true;
  }
  evt_set_last_tag("resunit_fwd<bf16, %d, k%d>", CI, KHP);
  hipLaunchKernelGGL((resunit_fwd<CI, NK>), dim3((int)blocks), dim3(256), lds, st, p);
  return evt_check_launch();
}

int nk_of(const evt_resunit_params* a) {
  if (a->C == 16) return ((a->k + 1) & ~1) * 16 / 32;     // prepared image pads the taps to an even count
  return a->k;                                             // C == 32: one tap per K step
}

}  // namespace

extern "C" {

int32_t evt_resunit_supported(const evt_resunit_params* a) {
  static const bool off = getenv("EVT_NO_RESUNIT") != nullptr;   // A/B switch for measurements
  if (off || !a || a->dtype != EVT_DT_BF16) return 0;
  if (a->C != 16 && a->C != 32) return 0;
  if (a->k != 3 && a->k != 7 && a->k != 11) return 0;
  if (a->dil < 1 || a->dil > 5 || a->nseq <= 0 || a->L < 64) return 0;
  return 1;
}

int evt_resunit_fwd(const evt_resunit_params* a, const void* x, const void* w1_reg, const void* w2_reg, const float* b1,
                    const float* b2, void* xa, void* mid_a, void* y, void* stream) {
  if (!evt_resunit_supported(a)) return EVT_ENOTSUP;
  if (!x || !w1_reg || !w2_reg || !y) return EVT_EINVAL;
  RUP p{};
  p.x = (const bf16_t*)x; p.w1 = (const bf16_t*)w1_reg; p.w2 = (const bf16_t*)w2_reg; p.b1 = b1; p.b2 = b2;
  p.xa = (bf16_t*)xa; p.mid = (bf16_t*)mid_a; p.y = (bf16_t*)y;
  p.nseq = a->nseq; p.L = a->L; p.k = a->k; p.dil = a->dil; p.slope = a->slope;
  hipStream_t st = (hipStream_t)stream;
  const int nk = nk_of(a);
  if (a->C == 16) {
    if (nk == 2) return launch<16, 2>(p, st);
    if (nk == 4) return launch<16, 4>(p, st);
    return launch<16, 6>(p, st);
  }
  if (nk == 3) return launch<32, 3>(p, st);
  if (nk == 7) return launch<32, 7>(p, st);
  return launch<32, 11>(p, st);
}

}  // extern "C"
