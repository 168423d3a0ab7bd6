// conv_narrow: stride-1 convolutions of the late HiFi-GAN stages (C = 16 and 32, k = 3 / 7 / 11, dilation 1 / 3 / 5;
// src/easevoice/module/modules.py:226-311 of the reference), forward and backward-data, gfx950 bf16.
//
// These layers move 21 MB per launch for 0.25-1.8 GMAC: they are the HBM-bound end of the vocoder, and the generic
// kernel spent its time on per-block overhead (weight staging, barriers, 24 MFMAs of work per wave).  Here:
//   * WEIGHTS STAY IN REGISTERS: a wave owns all output channels (16*MT) and keeps the whole [Cout][k*Cin] matrix as MFMA
//     A fragments (6..22 fragments), loaded once;
//   * waves are independent and persistent: each loops over 64-position units of the flat (sequence, tile) list, stages
//     its own input rows (+ dilated halo) through a private LDS region and never meets another wave at a barrier;
//   * the next unit's rows are prefetched into registers while the current unit is multiplied;
//   * im2col is an LDS view: B fragment of K step ks, lane (n, g) = 8 channels of row n + tap*dil (16-byte ds_read);
//   * epilogue: bias / activation / gate / residual, 8-byte stores that tile whole 32- or 64-byte position rows.
// The same kernel does backward-data with the tap-flipped transposed weight image (ALT), like conv_igemm.
#include "conv_p.h"

namespace evt_conv {
namespace {

// CI: K-side channels (16 or 32), MT: output channel tiles (Cout = 16*MT), NK: K steps = ceil(KHp*CI / 32)
template <int CI, int MT, int NK>
__global__ __launch_bounds__(256) void conv_narrow(ConvP p, int units_per_seq, long total_units, int region_rows) {
  constexpr int XROW = CI == 16 ? 32 : 96;              // LDS row pitch (bytes): 2 x odd 16-byte slots, conflict-free
  constexpr int PPR = CI * 2 / 16;                       // 16-byte pieces per input row
  constexpr int XPT = 8;                                 // prefetch registers per lane (region pieces <= 64 * XPT)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  unsigned char* xs = smem + wave * region_rows * XROW;
  const h16_t* X = reinterpret_cast<const h16_t*>(p.x);
  const h16_t* W = reinterpret_cast<const h16_t*>(p.w);

  // A fragments: row co = i*16 + n of the prepared image [co][1 chunk][KHp][CI] -> K index (tap, ci) is contiguous
  h16x8 af[MT][NK];
  const int ktot = p.KHp * CI;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
      const int kk = ks * 32 + g * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (kk < ktot) v = *reinterpret_cast<const uint4*>(W + (long)(i * 16 + n) * ktot + kk);
      union { uint4 u; h16x8 f; } c; c.u = v; af[i][ks] = c.f;
    }
  float bias[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) bias[i][r] = p.bias ? p.bias[i * 16 + g * 4 + r] : 0.f;

  const long wave_id = (long)blockIdx.x * 4 + wave, nwaves = (long)gridDim.x * 4;
  const int npieces = region_rows * PPR;
  uint4 xr[XPT];
  auto load_unit = [&](long u) {
    const int seq = (int)(u / units_per_seq);
    const int row0 = (int)(u - (long)seq * units_per_seq) * 64 + p.off_in;
    const h16_t* xg = X + (long)seq * p.Lin * CI;
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int idx = lane + i * 64;
      const int r = idx / PPR, part = idx - r * PPR;
      const int in_row = row0 + r;
      const bool ok = idx < npieces && in_row >= 0 && in_row < p.Lin;
      xr[i] = ok ? *reinterpret_cast<const uint4*>(xg + (long)in_row * CI + part * 8) : make_uint4(0, 0, 0, 0);
    }
  };
  if (wave_id < total_units) load_unit(wave_id);
  for (long u = wave_id; u < total_units; u += nwaves) {
    const int seq = (int)(u / units_per_seq);
    const int q0 = (int)(u - (long)seq * units_per_seq) * 64;
    // publish this unit's rows to the wave's LDS region (previous unit's fragment reads are complete: in-order DS)
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
      const int idx = lane + i * 64;
      if (idx < npieces) {
        const int r = idx / PPR, part = idx - r * PPR;
        *reinterpret_cast<uint4*>(xs + r * XROW + part * 16) = xr[i];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (u + nwaves < total_units) load_unit(u + nwaves);       // flies under the MFMAs below

    f32x4 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
      const int kk = ks * 32 + g * 8;
      const int tap = kk / CI, ci = kk - tap * CI;
      const unsigned char* base = xs + (n + tap * p.dil) * XROW + ci * 2;
      h16x8 b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const h16x8*>(base + j * 16 * XROW);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = EVT_MFMA_16x16x32(af[i][ks], b[j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();

    // epilogue: lane holds channels i*16 + g*4 .. +3 of position q0 + j*16 + n
    const long sbase = (long)seq * p.Lout * p.Cout;
    h16_t* yg = reinterpret_cast<h16_t*>(p.y) + sbase;
    const h16_t* rg = p.res ? reinterpret_cast<const h16_t*>(p.res) + sbase : nullptr;
    const h16_t* gg = p.gate ? reinterpret_cast<const h16_t*>(p.gate) + sbase : nullptr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = q0 + j * 16 + n;
      if (q >= p.Q) continue;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const long off = (long)q * p.Cout + i * 16 + g * 4;
        uint2 gv = make_uint2(0, 0), rv = make_uint2(0, 0);
        if (gg) gv = *reinterpret_cast<const uint2*>(gg + off);
        if (rg) rv = *reinterpret_cast<const uint2*>(rg + off);
        const h16_t* pg = reinterpret_cast<const h16_t*>(&gv);
        const h16_t* pr = reinterpret_cast<const h16_t*>(&rv);
        h16_t outv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[i][j][r] + bias[i][r];
          if (p.out_act == EVT_ACT_LRELU) v = lrelu_f(v, p.out_slope);
          else if (p.out_act == EVT_ACT_TANH) v = tanhf(v);
          if (gg) v *= (h2f(pg[r]) > 0.f ? 1.f : p.gate_slope);
          if (rg) v += h2f(pr[r]);
          outv[r] = f2h(v);
        }
        *reinterpret_cast<uint2*>(yg + off) = *reinterpret_cast<uint2*>(outv);
      }
    }
  }
}

template <int CI, int MT, int NK>
int launch_inst(const ConvP& p, hipStream_t st) {
  constexpr int XROW = CI == 16 ? 32 : 96;
  const int region_rows = 63 + (NK * 32 / CI - 1) * p.dil + 1;      // covers the zero-padded taps of the last K step
  if (region_rows * (CI * 2 / 16) > 64 * 8) return EVT_ENOTSUP;       // prefetch registers
  const size_t lds = (size_t)4 * region_rows * XROW;
  const int ups = (p.Q + 63) / 64;
  const long total = (long)p.nseq * ups;
  long blocks = (total + 3) / 4;
  static const long cap = getenv("EVT_NARROW_BLOCKS") ? atol(getenv("EVT_NARROW_BLOCKS")) : 1024;   // tuning knob
  if (blocks > cap) blocks = cap;                                      // resident blocks, persistent waves
  static bool attr = false;
  if (lds > 48 * 1024 && !attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_narrow<CI, MT, NK>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return EVT_ELAUNCH;
    attr = true;
  }
  evt_set_last_tag("conv_narrow<bf16, %d, %d, k%d>", CI, 16 * MT, NK * 32 / CI);
  hipLaunchKernelGGL((conv_narrow<CI, MT, NK>), dim3((int)blocks), dim3(256), lds, st, p, ups, total, region_rows);
  return evt_check_launch();
}

}  // namespace

bool narrow_eligible(const ConvP& p, int dtype, int out_ch, int k_ch, int nphase) {
  static const bool off = getenv("EVT_NO_NARROW") != nullptr;   // A/B switch for measurements
  if (off || dtype != EVT_DT_HALF || nphase != 1) return false;
  if (p.xact || p.in_slope != 1.f) return false;
  if (p.s_in != 1 || p.s_out != 1 || p.off_out != 0) return false;
  if (!((k_ch == 16 && out_ch == 16) || (k_ch == 32 && out_ch == 32))) return false;
  if (p.nchunk != 1) return false;
  const int nk = (p.KHp * k_ch + 31) / 32;
  if (k_ch == 16 && nk != 2 && nk != 4 && nk != 6) return false;   // k = 3(4) / 7(8) / 11(12) taps
  if (k_ch == 32 && nk != 3 && nk != 7) return false;              // k = 11 at C = 32 (88 fragment registers) measured slower
  const int region_rows = 63 + (nk * 32 / k_ch - 1) * p.dil + 1;
  if (region_rows * (k_ch * 2 / 16) > 64 * 8) return false;       // prefetch registers of one wave
  return (long)p.nseq * p.Q >= 4096;
}

int launch_conv_narrow(const ConvP& p, int out_ch, int k_ch, int nphase, hipStream_t st) {
  if (!narrow_eligible(p, EVT_DT_HALF, out_ch, k_ch, nphase)) return EVT_ENOTSUP;
  const int nk = (p.KHp * k_ch + 31) / 32;
  if (k_ch == 16) {
    if (nk == 2) return launch_inst<16, 1, 2>(p, st);
    if (nk == 4) return launch_inst<16, 1, 4>(p, st);
    return launch_inst<16, 1, 6>(p, st);
  }
  if (nk == 3) return launch_inst<32, 2, 3>(p, st);
  return launch_inst<32, 2, 7>(p, st);
}

}  // namespace evt_conv
