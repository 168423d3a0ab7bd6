// Conv1d / ConvTranspose1d family for gfx950: channels-last activations, implicit GEMM on MFMA.
//
// Reference call sites replaced (file:line under /root/reference):
//   HiFi-GAN Generator / ResBlock1      src/easevoice/module/models.py:452-471, modules.py:298-311
//   WN (posterior encoder, flow)        src/easevoice/module/modules.py:187-212
//   FFN convs                           src/easevoice/module/attentions.py:408-416
//   DiscriminatorS / DiscriminatorP     src/easevoice/module/models.py:538-587
//
// One kernel template (conv_igemm) serves forward, backward-data and ConvTranspose: the launch
// descriptor maps an output index q of a "unit" to input rows q*s_in + tap*dil + off_in and to the
// output row q*s_out + off_out (+phase).  Strided backward-data / ConvTranspose forward run as
// `stride` polyphase sub-convolutions (grid.y = phase).  GEMM view: M = output channels (MFMA rows),
// N = positions (MFMA columns), K = (tap, input channel) with the channel index contiguous in HBM
// and in LDS, so both MFMA operands are 16-byte ds_read_b128 fragments.
//
// Work decomposition: a wave owns one unit = 16*NT consecutive q of ONE sequence (never straddling a
// sequence) and all 16*MT output channels of the block's channel tile; the 4 waves of a block own 4
// consecutive units and share the weight tile staged in LDS.  blockIdx.x is decoded XCD-aware so the
// channel tiles that re-read the same activation rows sit on the same XCD (same L2).
#include "evt_common.h"
#include "../../include/evt.h"
#include "conv_p.h"
#include "wgrad_epi.h"

extern "C" int evt_grouped_supported(const evt_conv1d_params* c);
extern "C" int evt_grouped_fwd(const evt_conv1d_params* c, const void* x, const void* w_reg, const float* bias, void* y,
                               void* stream);
extern "C" int evt_grouped_bwd_data(const evt_conv1d_params* c, const void* dy, const void* y, const void* w_reg,
                                    void* dx, void* stream);
extern "C" int evt_grouped_bwd_weight(const evt_conv1d_params* c, const void* x, const void* dy, const void* y,
                                      float* dw, void* stream);

extern "C" int evt_small_kind(const evt_conv1d_params* c);
extern "C" int evt_cout1_fwd(const evt_conv1d_params* c, const void* x, const void* w_reg, const float* bias, void* y,
                             void* stream);
extern "C" int evt_cout1_bwd_weight(const evt_conv1d_params* c, const void* x, const void* dy, const void* y, float* dw,
                                    float* ws, long ws_floats, void* stream);
extern "C" int evt_cin1_bwd_weight(const evt_conv1d_params* c, const void* x, const void* dy, const void* y, float* dw,
                                   float* dbias, float* ws, long ws_floats, void* stream);
extern "C" int evt_cin1_fwd(const evt_conv1d_params* c, const void* x, const void* w_reg, const float* bias, void* y,
                            void* stream);
extern "C" int evt_cin1_bwd_data(const evt_conv1d_params* c, const void* dy, const void* y, const void* w_reg,
                                 const void* gate, const void* dx_add, void* dx, void* stream);
extern "C" int evt_cout1_bwd_data(const evt_conv1d_params* c, const void* dy, const void* y, const void* w_reg,
                                  const void* gate, const void* dx_add, void* dx, void* stream);

namespace {

using evt_conv::ConvP;

template <typename T> struct Frag;
template <> struct Frag<float> {
  static constexpr int EPL = 1, KS = 4;
  typedef float type;
  static __device__ __forceinline__ f32x4 mma(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
};
template <> struct Frag<h16_t> {
  static constexpr int EPL = 8, KS = 32;
  typedef h16x8 type;
  static __device__ __forceinline__ f32x4 mma(h16x8 a, h16x8 b, f32x4 c) {
    return EVT_MFMA_16x16x32(a, b, c, 0, 0, 0);
  }
};

// 16 bytes of T with the load-side fusion applied in fp32 and re-rounded to T.  MODE is resolved ONCE per staging call
// (a wave-uniform switch outside the piece loops) so the per-element code is branch-free:
//   0 copy | 1 leaky-relu(in_slope) | 2 x * lrelu'(act) | 3 x * tanh'(act) | 4 generic (both fusions)
template <typename T, int MODE>
__device__ __forceinline__ uint4 fuse16(uint4 v, uint4 va, float act_slope, float in_slope, int act_kind) {
  if (MODE == 0) return v;
  constexpr int V = 16 / sizeof(T);
  T* h = reinterpret_cast<T*>(&v);
  const T* a = reinterpret_cast<const T*>(&va);
#pragma unroll
  for (int i = 0; i < V; ++i) {
    float t = to_f<T>(h[i]);
    if (MODE == 1) t = t > 0.f ? t : t * in_slope;
    else if (MODE == 2) t = to_f<T>(a[i]) > 0.f ? t : t * act_slope;
    else if (MODE == 3) { const float y = to_f<T>(a[i]); t *= 1.f - y * y; }
    else { t *= dact_from_out(act_kind, to_f<T>(a[i]), act_slope); t = t > 0.f ? t : t * in_slope; }
    h[i] = from_f<T>(t);
  }
  return v;
}

__device__ __forceinline__ int fuse_mode(bool has_act, int act_kind, float in_slope) {
  if (!has_act) return in_slope == 1.f ? 0 : 1;
  if (in_slope != 1.f) return 4;
  return act_kind == EVT_ACT_LRELU ? 2 : (act_kind == EVT_ACT_TANH ? 3 : 0);
}

// legacy entry used by the weight-gradient kernels (mode resolved per call, still branch-free per element)
template <typename T>
__device__ __forceinline__ uint4 fuse_load16(uint4 v, bool has_act, uint4 va, int act_kind, float act_slope,
                                             float in_slope) {
  switch (fuse_mode(has_act, act_kind, in_slope)) {
    case 0: return v;
    case 1: return fuse16<T, 1>(v, va, act_slope, in_slope, act_kind);
    case 2: return fuse16<T, 2>(v, va, act_slope, in_slope, act_kind);
    case 3: return fuse16<T, 3>(v, va, act_slope, in_slope, act_kind);
    default: return fuse16<T, 4>(v, va, act_slope, in_slope, act_kind);
  }
}

// ---- software-pipeline helpers (kept as force-inlined functions with explicit array references: lambdas capturing
//      the register arrays defeated SROA in the largest instantiations and spilled the prefetch registers) ----
template <typename T, int CK, int WPT, int TPR>
__device__ __forceinline__ void igemm_load_w(uint4 (&wr)[WPT], const T* wg, const ConvP& p, int ch, int tg, int TG,
                                             int wrow, int wsub) {
  constexpr int SZ = sizeof(T);
  const int t0 = tg * TG;
  const int ppr = min(TG, p.KHp - t0) * CK * SZ / 16;
  const T* src = wg + ((long)(wrow * p.nchunk + ch) * p.KHp + t0) * CK;
#pragma unroll
  for (int i = 0; i < WPT; ++i) {
    const int piece = wsub + i * TPR;
    wr[i] = piece < ppr ? *reinterpret_cast<const uint4*>(src + piece * (16 / SZ)) : make_uint4(0, 0, 0, 0);
  }
}

template <typename T, int CK, int WPT, int TPR>
__device__ __forceinline__ void igemm_store_w(const uint4 (&wr)[WPT], unsigned char* ws, const ConvP& p, int tg, int TG,
                                              int WROW, int wrow, int wsub) {
  constexpr int SZ = sizeof(T);
  const int ppr = min(TG, p.KHp - tg * TG) * CK * SZ / 16;
#pragma unroll
  for (int i = 0; i < WPT; ++i) {
    const int piece = wsub + i * TPR;
    if (piece < ppr) *reinterpret_cast<uint4*>(ws + wrow * WROW + piece * 16) = wr[i];
  }
}

// region = a run of R rows of ONE sequence staged for one (or NT) 16-position tiles; a lane keeps XPR 16-byte pieces of
// it in xr[I0 .. I0+XPR)
template <typename T, int CK, int XPT, int I0, int XPR>
__device__ __forceinline__ void igemm_load_x(uint4 (&xr)[XPT], uint4 (&ar)[XPT], const T* xg, const T* ag,
                                             const ConvP& p, int ch, int R, int row0, int lane) {
  constexpr int SZ = sizeof(T), LPR = CK * SZ / 16;
#pragma unroll
  for (int i = 0; i < XPR; ++i) {
    const int idx = lane + i * 64;
    const int r = idx / LPR, part = idx - r * LPR;
    const int in_row = row0 + r;
    const bool ok = idx < R * LPR && in_row >= 0 && in_row < p.Lin;
    const long off = ok ? (long)in_row * p.Cin + ch * CK + part * (16 / SZ) : 0;
    xr[I0 + i] = ok ? *reinterpret_cast<const uint4*>(xg + off) : make_uint4(0, 0, 0, 0);
    ar[I0 + i] = (ok && ag) ? *reinterpret_cast<const uint4*>(ag + off) : make_uint4(0, 0, 0, 0);
  }
}

template <typename T, int CK, int XPT, int I0, int XPR, int MODE>
__device__ __forceinline__ void igemm_store_x_m(const uint4 (&xr)[XPT], const uint4 (&ar)[XPT], unsigned char* xs,
                                                const ConvP& p, int R, int row0, int lane, int XROW) {
  constexpr int SZ = sizeof(T), LPR = CK * SZ / 16;
#pragma unroll
  for (int i = 0; i < XPR; ++i) {
    const int idx = lane + i * 64;
    const int r = idx / LPR, part = idx - r * LPR;
    const int in_row = row0 + r;
    if (idx < R * LPR) {
      const bool ok = in_row >= 0 && in_row < p.Lin;
      uint4 v = fuse16<T, MODE>(xr[I0 + i], ar[I0 + i], p.xact_slope, p.in_slope, p.xact_kind);
      if (!ok) v = make_uint4(0, 0, 0, 0);
      *reinterpret_cast<uint4*>(xs + r * XROW + part * 16) = v;
    }
  }
}

template <typename T, int CK, int XPT, int I0, int XPR>
__device__ __forceinline__ void igemm_store_x(const uint4 (&xr)[XPT], const uint4 (&ar)[XPT], unsigned char* xs,
                                              int mode, const ConvP& p, int R, int row0, int lane, int XROW) {
  switch (mode) {
    case 0: igemm_store_x_m<T, CK, XPT, I0, XPR, 0>(xr, ar, xs, p, R, row0, lane, XROW); break;
    case 1: igemm_store_x_m<T, CK, XPT, I0, XPR, 1>(xr, ar, xs, p, R, row0, lane, XROW); break;
    case 2: igemm_store_x_m<T, CK, XPT, I0, XPR, 2>(xr, ar, xs, p, R, row0, lane, XROW); break;
    case 3: igemm_store_x_m<T, CK, XPT, I0, XPR, 3>(xr, ar, xs, p, R, row0, lane, XROW); break;
    default: igemm_store_x_m<T, CK, XPT, I0, XPR, 4>(xr, ar, xs, p, R, row0, lane, XROW); break;
  }
}

template <typename T, int CK, int MODE>
__device__ __forceinline__ void igemm_stage_x_sync_m(unsigned char* xs, const T* xg, const T* ag, const ConvP& p, int ch,
                                                     int R, int row0, int lane, int XROW, int first_idx) {
  constexpr int SZ = sizeof(T), LPR = CK * SZ / 16;
  for (int idx = first_idx + lane; idx < R * LPR; idx += 64) {
    const int r = idx / LPR, part = idx - r * LPR;
    const int in_row = row0 + r;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (in_row >= 0 && in_row < p.Lin) {
      const long off = (long)in_row * p.Cin + ch * CK + part * (16 / SZ);
      v = *reinterpret_cast<const uint4*>(xg + off);
      uint4 va = make_uint4(0, 0, 0, 0);
      if (MODE >= 2) va = *reinterpret_cast<const uint4*>(ag + off);
      v = fuse16<T, MODE>(v, va, p.xact_slope, p.in_slope, p.xact_kind);
    }
    *reinterpret_cast<uint4*>(xs + r * XROW + part * 16) = v;
  }
}

template <typename T, int CK>
__device__ __forceinline__ void igemm_stage_x_sync(unsigned char* xs, const T* xg, const T* ag, int mode, const ConvP& p,
                                                   int ch, int R, int row0, int lane, int XROW, int first_idx) {
  switch (mode) {
    case 0: igemm_stage_x_sync_m<T, CK, 0>(xs, xg, ag, p, ch, R, row0, lane, XROW, first_idx); break;
    case 1: igemm_stage_x_sync_m<T, CK, 1>(xs, xg, ag, p, ch, R, row0, lane, XROW, first_idx); break;
    case 2: igemm_stage_x_sync_m<T, CK, 2>(xs, xg, ag, p, ch, R, row0, lane, XROW, first_idx); break;
    case 3: igemm_stage_x_sync_m<T, CK, 3>(xs, xg, ag, p, ch, R, row0, lane, XROW, first_idx); break;
    default: igemm_stage_x_sync_m<T, CK, 4>(xs, xg, ag, p, ch, R, row0, lane, XROW, first_idx); break;
  }
}

// region loop helper: compile-time recursion over the (at most 4) regions of a wave
template <typename T, int CK, int XPT, int XPR, int NREG, int J>
struct RegionOps {
  static __device__ __forceinline__ void load(uint4 (&xr)[XPT], uint4 (&ar)[XPT], const T* const (&xg)[NREG],
                                              const T* const (&ag)[NREG], const bool (&ract)[NREG], const ConvP& p,
                                              int ch, int Rr, const int (&rrow0)[NREG], int lane) {
    if (ract[J]) igemm_load_x<T, CK, XPT, J * XPR, XPR>(xr, ar, xg[J], ag[J], p, ch, Rr, rrow0[J], lane);
    if constexpr (J + 1 < NREG) RegionOps<T, CK, XPT, XPR, NREG, J + 1>::load(xr, ar, xg, ag, ract, p, ch, Rr, rrow0, lane);
  }
  static __device__ __forceinline__ void store(const uint4 (&xr)[XPT], const uint4 (&ar)[XPT], unsigned char* xs,
                                               const T* const (&xg)[NREG], const T* const (&ag)[NREG],
                                               const bool (&ract)[NREG], int mode, const ConvP& p, int ch, int Rr,
                                               const int (&rrow0)[NREG], int lane, int XROW, bool prefetched) {
    constexpr int LPR = CK * sizeof(T) / 16;
    if (ract[J]) {
      unsigned char* base = xs + J * Rr * XROW;
      if (prefetched) {
        igemm_store_x<T, CK, XPT, J * XPR, XPR>(xr, ar, base, mode, p, Rr, rrow0[J], lane, XROW);
        if (Rr * LPR > 64 * XPR) igemm_stage_x_sync<T, CK>(base, xg[J], ag[J], mode, p, ch, Rr, rrow0[J], lane, XROW, 64 * XPR);
      } else {
        igemm_stage_x_sync<T, CK>(base, xg[J], ag[J], mode, p, ch, Rr, rrow0[J], lane, XROW, 0);
      }
    }
    if constexpr (J + 1 < NREG)
      RegionOps<T, CK, XPT, XPR, NREG, J + 1>::store(xr, ar, xs, xg, ag, ract, mode, p, ch, Rr, rrow0, lane, XROW, prefetched);
  }
};

// SPLIT = false: a wave owns 16*NT CONSECUTIVE positions of one sequence (one staged region, halo shared).
// SPLIT = true : a wave owns NT independent 16-position units that may belong to different sequences (NT regions);
//                short sequences (DiscriminatorP with period 5/7/11: 23..51 positions) keep the 4x4 register tiling.
template <typename T, int CK, int MT, int NT, bool PF, bool SPLIT>
__global__ __launch_bounds__(256) void conv_igemm(ConvP p) {
  constexpr int EPL = Frag<T>::EPL, KS = Frag<T>::KS;
  constexpr int TM = 16 * MT, PW = 16 * NT;
  constexpr int SZ = sizeof(T);
  // row pitches in 16-byte slots are 2 x odd: ds_read_b128 lane groups then hit 16 distinct slots (no bank conflict)
  constexpr int XROW = SZ == 4 ? CK * SZ + 16 : (CK == 16 ? 32 : 96);   // fp32 (ds_read_b32) keeps the +16 pitch
  constexpr int WK = (SZ == 2 ? 256 : 128);      // K elements per weight stage
  constexpr int TG = WK / CK;                    // taps per weight stage
  constexpr int WROW = WK * SZ + (SZ == 4 ? 16 : 32);  // bytes per staged weight row (bf16: 34 slots)
  constexpr int NREG = SPLIT ? NT : 1;
  typedef typename Frag<T>::type frag_t;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g = lane >> 4;

  // XCD-aware decode: consecutive blockIdx.x round-robin over 8 XCDs; give every XCD whole
  // position blocks (all Y channel tiles of a position block land on one XCD / one L2).
  const int lin = blockIdx.x;
  const int xcd = lin & 7, slot = lin >> 3;
  const int yi = slot % p.Y;
  const int pb = xcd + 8 * (slot / p.Y);
  if (pb >= p.P) return;
  const int phase = blockIdx.y;

  // rows staged per region
  const int Rr = ((SPLIT ? 16 : PW) - 1) * p.s_in + (p.KHp - 1) * p.dil + 1;
  unsigned char* ws = smem;
  unsigned char* xs = smem + TM * WROW + wave * (NREG * Rr * XROW);

  // units: p.U units per sequence, each SPLIT ? 16 : PW positions
  bool ract[NREG];
  int rseq[NREG], rq0[NREG], rrow0[NREG];
  const T* xg[NREG];
  const T* ag[NREG];
  bool active = false;
#pragma unroll
  for (int j = 0; j < NREG; ++j) {
    const long u = ((long)pb * 4 + wave) * NREG + j;
    ract[j] = u < (long)p.nseq * p.U;
    rseq[j] = ract[j] ? (int)(u / p.U) : 0;
    rq0[j] = ract[j] ? (int)(u % p.U) * (SPLIT ? 16 : PW) : 0;
    rrow0[j] = rq0[j] * p.s_in + p.off_in;
    xg[j] = reinterpret_cast<const T*>(p.x) + (long)rseq[j] * p.Lin * p.Cin;
    ag[j] = p.xact ? reinterpret_cast<const T*>(p.xact) + (long)rseq[j] * p.Lin * p.Cin : nullptr;
    active = active || ract[j];
  }
  const T* wg = reinterpret_cast<const T*>(p.w) + (long)phase * p.w_phase_stride +
                (long)yi * TM * p.nchunk * p.KHp * CK;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Software pipeline (register prefetch): the global loads of stage s+1 are issued before the MFMAs of stage s and
  // only written to LDS after the next barrier, so HBM/L2 latency hides behind compute.  The fused load-side
  // arithmetic (leaky-relu / activation derivative) is applied at LDS-write time so nothing waits on a load early.
  constexpr int TPR = 256 / TM;                 // threads per weight row
  constexpr int WPT = (WK * SZ / 16) / TPR;     // 16-byte weight pieces per thread per stage (= 2*MT)
  constexpr int XPT = PF ? (SZ == 2 ? 8 : 12) : NREG;  // 16-byte activation pieces per lane held in registers
  constexpr int XPR = XPT / NREG;               // ... per region
  const int ngroups = (p.KHp + TG - 1) / TG;
  const int nst = p.nchunk * ngroups;
  uint4 wr[WPT], xr[XPT], ar[XPT];
  const int wrow = tid / TPR, wsub = tid % TPR;
  const int fmode = fuse_mode(p.xact != nullptr, p.xact_kind, p.in_slope);
  typedef RegionOps<T, CK, XPT, XPR, NREG, 0> RO;

  if (PF) {
    igemm_load_w<T, CK, WPT, TPR>(wr, wg, p, 0, 0, TG, wrow, wsub);
    RO::load(xr, ar, xg, ag, ract, p, 0, Rr, rrow0, lane);
  }
  for (int st = 0; st < nst; ++st) {
    const int ch = st / ngroups, tg = st - ch * ngroups;
    __syncthreads();  // previous stage's fragment reads are done
    if (tg == 0) RO::store(xr, ar, xs, xg, ag, ract, fmode, p, ch, Rr, rrow0, lane, XROW, PF);
    if (!PF) igemm_load_w<T, CK, WPT, TPR>(wr, wg, p, ch, tg, TG, wrow, wsub);
    igemm_store_w<T, CK, WPT, TPR>(wr, ws, p, tg, TG, WROW, wrow, wsub);
    __syncthreads();
    if (PF && st + 1 < nst) {
      const int nch = (st + 1) / ngroups, ntg = (st + 1) - nch * ngroups;
      igemm_load_w<T, CK, WPT, TPR>(wr, wg, p, nch, ntg, TG, wrow, wsub);
      if (ntg == 0) RO::load(xr, ar, xg, ag, ract, p, nch, Rr, rrow0, lane);
    }
    if (active) {
      const int t0 = tg * TG;
      const int ntap = min(TG, p.KHp - t0);
      const int steps = ntap * CK / KS;
      for (int k = 0; k < steps; ++k) {
        const int kl = k * KS + g * EPL;  // K index inside this weight stage: [tap][ci]
        const int tl = kl / CK, ci = kl - tl * CK;
        const int tap = t0 + tl;
        frag_t a[MT], b[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
          a[i] = *reinterpret_cast<const frag_t*>(ws + (i * 16 + n) * WROW + kl * SZ);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int rowj = SPLIT ? j * Rr + n * p.s_in : (j * 16 + n) * p.s_in;
          b[j] = *reinterpret_cast<const frag_t*>(xs + (rowj + tap * p.dil) * XROW + ci * SZ);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] = Frag<T>::mma(a[i], b[j], acc[i][j]);
      }
    }
  }
  if (!active) return;

  // epilogue: lane holds rows (channels) g*4..g*4+3, column (position) n of each 16x16 tile
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int rj = SPLIT ? j : 0;
    const int q = SPLIT ? rq0[rj] + n : rq0[0] + j * 16 + n;
    const int orow = q * p.s_out + p.off_out + phase * p.off_out_phase;
    if (!ract[rj] || q >= p.Q || orow < 0 || orow >= p.Lout) continue;
    const long sbase = (long)rseq[rj] * p.Lout * p.Cout;
    T* yg = reinterpret_cast<T*>(p.y) + sbase;
    const T* rg = p.res ? reinterpret_cast<const T*>(p.res) + sbase : nullptr;
    const T* gg = p.gate ? reinterpret_cast<const T*>(p.gate) + sbase : nullptr;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int co = yi * TM + i * 16 + g * 4;
      const long off = (long)orow * p.Cout + co;
      T outv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[i][j][r];
        if (p.bias) v += p.bias[co + r];
        if (p.out_act == EVT_ACT_LRELU) v = lrelu_f(v, p.out_slope);
        else if (p.out_act == EVT_ACT_TANH) v = tanhf(v);
        if (gg) v *= (to_f<T>(gg[off + r]) > 0.f ? 1.f : p.gate_slope);
        if (rg) v += to_f<T>(rg[off + r]);
        outv[r] = from_f<T>(v);
      }
      if constexpr (SZ == 4) *reinterpret_cast<float4*>(yg + off) = *reinterpret_cast<float4*>(outv);
      else *reinterpret_cast<uint2*>(yg + off) = *reinterpret_cast<uint2*>(outv);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Weight gradient: dW[a][chunk(b)][tap][cc] += sum_{seq,q} A[seq][q][a] * B[seq][q*s + tap*dil + off][b]
// GEMM view: M = A channels, N = B channels (one CK chunk), K = positions.  A block owns 64 A-channels
// (16 per wave), one B chunk and up to KT taps; positions are split over blockIdx.y and the partial
// tiles are accumulated with fp32 global atomics.  Fragments are gathered element-wise from
// position-major LDS tiles (k = position is the strided index of a channels-last tensor).
// ---------------------------------------------------------------------------------------------
using evt_conv::WgP;

template <typename T> union FragBuf;
template <> union FragBuf<float> { float v; float e[1]; };
template <> union FragBuf<h16_t> { h16x8 v; h16_t e[8]; };

template <typename T>
__device__ __forceinline__ T fuse_elem(T v, bool has_act, T va, int kind, float aslope, float slope) {
  if (!has_act && slope == 1.f) return v;
  float t = to_f<T>(v);
  if (has_act) t *= dact_from_out(kind, to_f<T>(va), aslope);
  return from_f<T>(lrelu_f(t, slope));
}

template <typename T, int CK>
__global__ __launch_bounds__(256) void conv_wgrad(WgP p) {
  constexpr int EPL = Frag<T>::EPL, KS = Frag<T>::KS;
  constexpr int KT = 4;    // taps per block
  constexpr int PK = 64;   // positions staged per iteration
  constexpr int NTB = CK / 16;
  constexpr int AP = 64 + 2;  // A tile pitch (elements): [PK][64 ch]
  constexpr int BP = CK + 2;  // B tile pitch (elements)
  typedef typename Frag<T>::type frag_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g = lane >> 4;

  // blockIdx.x = ((atile * nchunk) + chunk) * ntapgrp + tapgrp
  int bx = blockIdx.x;
  const int tgi = bx % p.ntapgrp; bx /= p.ntapgrp;
  const int ch = bx % p.nchunk;
  const int atile = bx / p.nchunk;
  const int t0 = tgi * KT;
  const int ntap = min(KT, p.KHp - t0);
  const int a0 = atile * 64;

  const int BR = (PK - 1) * p.s + (KT - 1) * p.dil + 1;  // staged B rows
  T* As = reinterpret_cast<T*>(smem);
  T* Bs = As + PK * AP;

  f32x4 acc[KT][NTB];
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int j = 0; j < NTB; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int UQ = (p.Q + PK - 1) / PK;
  const long total = (long)p.nseq * UQ;
  const T* Ag0 = reinterpret_cast<const T*>(p.A);
  const T* Aa0 = reinterpret_cast<const T*>(p.Aact);
  const T* Bg0 = reinterpret_cast<const T*>(p.B);
  const T* Ba0 = reinterpret_cast<const T*>(p.Bact);

  for (long it = blockIdx.y; it < total; it += p.nsplit) {
    const int seq = (int)(it / UQ);
    const int q0 = (int)(it % UQ) * PK;
    __syncthreads();
    // stage A tile: PK positions x 64 channels
    for (int idx = tid; idx < PK * 64; idx += 256) {
      const int r = idx >> 6, c = idx & 63;
      const int q = q0 + r;
      T v = from_f<T>(0.f);
      if (q < p.Q && a0 + c < p.CA) {
        const long off = ((long)seq * p.LA + q) * p.CA + a0 + c;
        v = Ag0[off];
        T va = Aa0 ? Aa0[off] : v;
        v = fuse_elem<T>(v, Aa0 != nullptr, va, p.aact_kind, p.aact_slope, p.a_slope);
      }
      As[r * AP + c] = v;
    }
    // stage B tile: rows q0*s + t0*dil + off + [0, BR)
    const int brow0 = q0 * p.s + t0 * p.dil + p.off;
    for (int idx = tid; idx < BR * CK; idx += 256) {
      const int r = idx / CK, c = idx - r * CK;
      const int row = brow0 + r;
      T v = from_f<T>(0.f);
      if (row >= 0 && row < p.LB) {
        const long off = ((long)seq * p.LB + row) * p.CB + ch * CK + c;
        v = Bg0[off];
        T va = Ba0 ? Ba0[off] : v;
        v = fuse_elem<T>(v, Ba0 != nullptr, va, p.bact_kind, p.bact_slope, p.b_slope);
      }
      Bs[r * BP + c] = v;
    }
    __syncthreads();
#pragma unroll 1
    for (int kk = 0; kk < PK / KS; ++kk) {
      const int k0 = kk * KS + g * EPL;  // first of this lane's EPL positions
      FragBuf<T> av;
#pragma unroll
      for (int e = 0; e < EPL; ++e) av.e[e] = As[(k0 + e) * AP + wave * 16 + n];
      const frag_t a = av.v;
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        if (t < ntap) {
#pragma unroll
          for (int j = 0; j < NTB; ++j) {
            FragBuf<T> bv;
#pragma unroll
            for (int e = 0; e < EPL; ++e) bv.e[e] = Bs[((k0 + e) * p.s + t * p.dil) * BP + j * 16 + n];
            acc[t][j] = Frag<T>::mma(a, bv.v, acc[t][j]);
          }
        }
      }
    }
  }
  // accumulate: lane holds A-channel rows g*4+r (of this wave's 16), B-channel column n
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    if (t >= ntap) continue;
#pragma unroll
    for (int j = 0; j < NTB; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int a = a0 + wave * 16 + g * 4 + r;
        if (a < p.CA) {
          const long off = (((long)a * p.nchunk + ch) * p.KHp + t0 + t) * CK + j * 16 + n;
          atomicAdd(p.dw + off, acc[t][j][r]);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// bf16 weight gradient, v2.  Same GEMM view as conv_wgrad (M = A channels, N = B channels, K = positions) but
//   * tiles are staged with 16-byte loads and kept position-major ([pos][channel]) in LDS;
//   * MFMA fragments (8 consecutive positions of one channel) come from ds_read_b64_tr_b16, the gfx950 LDS
//     transpose read: inside a 16-lane group lane j supplies the address of 4 contiguous elements of row (j>>2),
//     columns 4*(j&3).., and lane i receives column i of that 4x16 block (semantics checked on hardware with
//     tools/probe_tr.hip).  The row address is per lane, so strided / tap-shifted rows cost nothing;
//   * for TA < 64 output-channel tiles the spare waves split the positions instead of idling;
//   * dbias (column sums of dy_eff) is accumulated from the staged A tile by the (chunk 0, tap group 0) blocks.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ h16x8 lds_tr2(const h16_t* p0, const h16_t* p1) {
  const unsigned a0 = (unsigned)(uintptr_t)p0, a1 = (unsigned)(uintptr_t)p1;
  uint2 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(lo), "=&v"(hi)
               : "v"(a0), "v"(a1)
               : "memory");
  union { uint4 u; h16x8 v; } r;
  r.u = make_uint4(lo.x, lo.y, hi.x, hi.y);
  return r.v;
}

// four fragments (eight transpose reads) behind ONE wait: the MFMAs that consume them then issue back to back
__device__ __forceinline__ void lds_tr2x4(const h16_t* p0, const h16_t* p1, const h16_t* p2, const h16_t* p3, int hi_off,
                                          h16x8& f0, h16x8& f1, h16x8& f2, h16x8& f3) {
  const unsigned a0 = (unsigned)(uintptr_t)p0, a1 = (unsigned)(uintptr_t)p1, a2 = (unsigned)(uintptr_t)p2,
                 a3 = (unsigned)(uintptr_t)p3;
  const unsigned b0 = a0 + hi_off, b1 = a1 + hi_off, b2 = a2 + hi_off, b3 = a3 + hi_off;
  uint2 l0, h0, l1, h1, l2, h2, l3, h3;
  asm volatile(
      "ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %9\n\t"
      "ds_read_b64_tr_b16 %2, %10\n\tds_read_b64_tr_b16 %3, %11\n\t"
      "ds_read_b64_tr_b16 %4, %12\n\tds_read_b64_tr_b16 %5, %13\n\t"
      "ds_read_b64_tr_b16 %6, %14\n\tds_read_b64_tr_b16 %7, %15\n\ts_waitcnt lgkmcnt(0)"
      : "=&v"(l0), "=&v"(h0), "=&v"(l1), "=&v"(h1), "=&v"(l2), "=&v"(h2), "=&v"(l3), "=&v"(h3)
      : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3)
      : "memory");
  union { uint4 u; h16x8 v; } r;
  r.u = make_uint4(l0.x, l0.y, h0.x, h0.y); f0 = r.v;
  r.u = make_uint4(l1.x, l1.y, h1.x, h1.y); f1 = r.v;
  r.u = make_uint4(l2.x, l2.y, h2.x, h2.y); f2 = r.v;
  r.u = make_uint4(l3.x, l3.y, h3.x, h3.y); f3 = r.v;
}

template <int CK, int TA>
__global__ __launch_bounds__(256) void conv_wgrad_tr(WgP p) {
  typedef h16_t T;
  constexpr int KT = 4;
  constexpr int NCT = TA / 16;   // waves along the A channels
  constexpr int NPS = 4 / NCT;   // waves along the positions
  constexpr int PK = 64 * NPS;   // positions staged per iteration (each wave: 64 = 2 MFMA k-steps)
  constexpr int NTB = CK / 16;
  constexpr int PA = TA + 8, PB = CK + 8;  // LDS pitches in elements (16-byte aligned rows)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j16 = lane & 15, g8 = lane >> 4;
  const int ct = wave % NCT, ps = wave / NCT;

  int bx = blockIdx.x;
  const int tgi = bx % p.ntapgrp; bx /= p.ntapgrp;
  const int ch = bx % p.nchunk;
  const int atile = bx / p.nchunk;
  const int t0 = tgi * KT;
  const int ntap = min(KT, p.KHp - t0);
  const int a0 = atile * TA;
  const int BR = (PK - 1) * p.s + (KT - 1) * p.dil + 1;
  T* As = reinterpret_cast<T*>(smem);
  T* Bs = As + PK * PA;

  f32x4 acc[KT][NTB];
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int j = 0; j < NTB; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const bool do_bias = p.dbias != nullptr && ch == 0 && tgi == 0;
  const int bcol = tid % TA, brow = tid / TA;   // dbias: thread sums column bcol over rows brow, brow + 256/TA, ...
  float bsum = 0.f;

  const int UQ = (p.Q + PK - 1) / PK;
  const long total = (long)p.nseq * UQ;
  const T* Ag0 = reinterpret_cast<const T*>(p.A);
  const T* Aa0 = reinterpret_cast<const T*>(p.Aact);
  const T* Bg0 = reinterpret_cast<const T*>(p.B);
  const T* Ba0 = reinterpret_cast<const T*>(p.Bact);

  for (long it = blockIdx.y; it < total; it += p.nsplit) {
    const int seq = (int)(it / UQ);
    const int q0 = (int)(it % UQ) * PK;
    __syncthreads();
    for (int idx = tid; idx < PK * (TA / 8); idx += 256) {
      const int r = idx / (TA / 8), c8 = idx - r * (TA / 8);
      uint4 v = make_uint4(0, 0, 0, 0);
      if (q0 + r < p.Q) {
        const long off = ((long)seq * p.LA + q0 + r) * p.CA + a0 + c8 * 8;
        v = *reinterpret_cast<const uint4*>(Ag0 + off);
        uint4 va = make_uint4(0, 0, 0, 0);
        if (Aa0) va = *reinterpret_cast<const uint4*>(Aa0 + off);
        v = fuse_load16<T>(v, Aa0 != nullptr, va, p.aact_kind, p.aact_slope, p.a_slope);
      }
      *reinterpret_cast<uint4*>(As + r * PA + c8 * 8) = v;
    }
    const int brow0 = q0 * p.s + t0 * p.dil + p.off;
    for (int idx = tid; idx < BR * (CK / 8); idx += 256) {
      const int r = idx / (CK / 8), c8 = idx - r * (CK / 8);
      const int row = brow0 + r;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (row >= 0 && row < p.LB) {
        const long off = ((long)seq * p.LB + row) * p.CB + ch * CK + c8 * 8;
        v = *reinterpret_cast<const uint4*>(Bg0 + off);
        uint4 va = make_uint4(0, 0, 0, 0);
        if (Ba0) va = *reinterpret_cast<const uint4*>(Ba0 + off);
        v = fuse_load16<T>(v, Ba0 != nullptr, va, p.bact_kind, p.bact_slope, p.b_slope);
      }
      *reinterpret_cast<uint4*>(Bs + r * PB + c8 * 8) = v;
    }
    __syncthreads();
    if (do_bias)
      for (int r = brow; r < PK; r += 256 / TA) bsum += h2f(As[r * PA + bcol]);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int kb = ps * 64 + ks * 32 + g8 * 8;  // first of this lane group's 8 positions
      const T* pa = As + (kb + (j16 >> 2)) * PA + ct * 16 + 4 * (j16 & 3);
      const h16x8 a = lds_tr2(pa, pa + 4 * PA);
      // taps beyond ntap read rows that are staged (the B tile always covers KT taps) and are simply not accumulated
      const T* pb0 = Bs + ((kb + (j16 >> 2)) * p.s) * PB + 4 * (j16 & 3);
      const int hi = 4 * p.s * PB * (int)sizeof(T);
#pragma unroll
      for (int j = 0; j < NTB; ++j) {
        h16x8 b0, b1, b2, b3;
        lds_tr2x4(pb0 + j * 16, pb0 + p.dil * PB + j * 16, pb0 + 2 * p.dil * PB + j * 16,
                  pb0 + 3 * p.dil * PB + j * 16, hi, b0, b1, b2, b3);
        acc[0][j] = EVT_MFMA_16x16x32(a, b0, acc[0][j], 0, 0, 0);
        if (ntap > 1) acc[1][j] = EVT_MFMA_16x16x32(a, b1, acc[1][j], 0, 0, 0);
        if (ntap > 2) acc[2][j] = EVT_MFMA_16x16x32(a, b2, acc[2][j], 0, 0, 0);
        if (ntap > 3) acc[3][j] = EVT_MFMA_16x16x32(a, b3, acc[3][j], 0, 0, 0);
      }
    }
  }
  if (do_bias) {
    // one atomic per column and block: same-address fp32 atomics from different XCDs serialise at the memory side
    // (~50 ns each), so 256 per block made this the whole cost of the narrow layers
    float* bred = reinterpret_cast<float*>(smem);
    __syncthreads();
    bred[tid] = bsum;
    __syncthreads();
    if (tid < TA) {
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 256 / TA; ++r) sum += bred[r * TA + tid];
      if (p.parts > 0) evt_conv::wg_finish_bias(p, a0 + tid, sum, blockIdx.y);
      else if (p.ws) p.ws[(long)p.nsplit * ((long)p.CA * p.nchunk * p.KHp * CK) + (long)blockIdx.y * p.CA + a0 + tid] = sum;
      else atomicAdd(p.dbias + a0 + tid, sum);
    }
  }
  if (p.parts > 0 && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) p.used[0] = p.now_used;
  if (NPS > 1) {
    // sum the position-slice waves inside the block first: 1/NPS of the global atomics
    float* red = reinterpret_cast<float*>(smem);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
      for (int j = 0; j < NTB; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(((wave * KT + t) * NTB + j) * 4 + r) * 64 + lane] = acc[t][j][r];
    __syncthreads();
    if (ps != 0) return;
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
      for (int j = 0; j < NTB; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[t][j][r];
          for (int o = 1; o < NPS; ++o) v += red[((((o * NCT + ct) * KT + t) * NTB + j) * 4 + r) * 64 + lane];
          acc[t][j][r] = v;
        }
  }
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    if (t >= ntap) continue;
#pragma unroll
    for (int j = 0; j < NTB; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int a = a0 + ct * 16 + g8 * 4 + r;
        const long off = (((long)a * p.nchunk + ch) * p.KHp + t0 + t) * CK + j * 16 + j16;
        // slab of this position split (evt_wn_grad_multi adds the slabs in order), or a scratch row (fold.hip does), or
        // one atomic per split and address
        if (p.parts > 0) {
          float* d = evt_conv::wg_slab(p, blockIdx.y) + off;
          const bool add = blockIdx.y == 0 ? p.dirty0 != 0 : (int)blockIdx.y < p.prev_used;
          *d = add ? *d + acc[t][j][r] : acc[t][j][r];
        } else if (p.ws) p.ws[(long)blockIdx.y * ((long)p.CA * p.nchunk * p.KHp * CK) + off] = acc[t][j][r];
        else atomicAdd(p.dw + off, acc[t][j][r]);
      }
    }
  }
}

// dbias[c] += sum over rows of dy * act'(y); [rows][C] channels-last.  A thread owns one 16-byte column group and walks
// the block's rows with register accumulators (rows_par rows in flight per pass); one LDS merge and one global atomic per
// channel and block.  Requires C % V == 0 and C / V <= 256.
template <typename T>
__global__ __launch_bounds__(256) void colsum_act2(const T* dy, const T* ys, float* out, long rows, int C, int kind,
                                                   float slope, int rows_per_block, float* ws) {
  __shared__ float acc[2048];                 // [rows_par][C]: rows_par * C = (256 / ppr) * ppr * V <= 2048
  constexpr int V = 16 / sizeof(T);
  const int ppr = C / V;
  const int rows_par = 256 / ppr;
  const int piece = threadIdx.x % ppr, rsub = threadIdx.x / ppr;
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = min(rows, r0 + rows_per_block);
  float a[V];
#pragma unroll
  for (int e = 0; e < V; ++e) a[e] = 0.f;
  if (rsub < rows_par) {
    for (long r = r0 + rsub; r < r1; r += rows_par) {
      const long off = r * C + piece * V;
      const uint4 v = *reinterpret_cast<const uint4*>(dy + off);
      uint4 va = make_uint4(0, 0, 0, 0);
      if (ys) va = *reinterpret_cast<const uint4*>(ys + off);
      const T* pv = reinterpret_cast<const T*>(&v);
      const T* pa = reinterpret_cast<const T*>(&va);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        float d = to_f<T>(pv[e]);
        if (ys) d *= dact_from_out(kind, to_f<T>(pa[e]), slope);
        a[e] += d;
      }
    }
#pragma unroll
    for (int e = 0; e < V; ++e) acc[rsub * C + piece * V + e] = a[e];
  }
  __syncthreads();
  // the row lanes are added in lane order; the block's sums go to a scratch row (fold.hip) or, without one, to atomics
  for (int c = threadIdx.x; c < C; c += 256) {
    float v = acc[c];
    for (int l = 1; l < rows_par; ++l) v += acc[l * C + c];
    if (ws) ws[(long)blockIdx.x * C + c] = v;
    else atomicAdd(out + c, v);
  }
}

// ---------------------------------------------------------------------------------------------
// Direct-form kernels: any channel count, groups, any taps.  They read the REG weight image only.
// Used for the degenerate shapes (Cin == 1, Cout == 1, grouped DiscriminatorS layers) and as the
// independent second implementation the parity tests compare the MFMA path against.
// ---------------------------------------------------------------------------------------------
struct NvP {
  const void* x; const void* w; const float* bias; const void* res; const void* y_in; const void* gate;
  void* y; float* dw; float* dbias;
  int nseq, lin, lout, cin, cout, k, stride, pad, dil, groups, transposed;
  int ck, nchunk, kp;    // REG geometry of the weight image [d0][nchunk][kp][ck]
  float in_slope; int out_act; float out_slope;
  int nsplit;
};

__device__ __forceinline__ long reg_index(const NvP& p, int d0, int d1, int t) {
  const int chunk = d1 / p.ck, cc = d1 - chunk * p.ck;
  return (((long)d0 * p.nchunk + chunk) * p.kp + t) * p.ck + cc;
}

// forward: one thread per output element (seq, row, co)
template <typename T>
__global__ void conv_naive_fwd(NvP p) {
  const long total = (long)p.nseq * p.lout * p.cout;
  const T* x = reinterpret_cast<const T*>(p.x);
  const T* w = reinterpret_cast<const T*>(p.w);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int co = (int)(i % p.cout);
    const long sr = i / p.cout;
    const int row = (int)(sr % p.lout);
    const int seq = (int)(sr / p.lout);
    float acc = 0.f;
    if (!p.transposed) {
      const int cpg = p.cin / p.groups, opg = p.cout / p.groups;
      const int grp = co / opg;
      for (int t = 0; t < p.k; ++t) {
        const int ir = row * p.stride + t * p.dil - p.pad;
        if (ir < 0 || ir >= p.lin) continue;
        const T* xr = x + ((long)seq * p.lin + ir) * p.cin + grp * cpg;
        for (int c = 0; c < cpg; ++c)
          acc += lrelu_f(to_f<T>(xr[c]), p.in_slope) * to_f<T>(w[reg_index(p, co, c, t)]);
      }
    } else {
      // out[n][co] = sum_{t,ci} x[(n + pad - t)/s][ci] * W[ci][co][t]
      for (int t = 0; t < p.k; ++t) {
        const int num = row + p.pad - t * p.dil;
        if (num < 0 || num % p.stride) continue;
        const int ir = num / p.stride;
        if (ir >= p.lin) continue;
        const T* xr = x + ((long)seq * p.lin + ir) * p.cin;
        for (int c = 0; c < p.cin; ++c)
          acc += lrelu_f(to_f<T>(xr[c]), p.in_slope) * to_f<T>(w[reg_index(p, c, co, t)]);
      }
    }
    if (p.bias) acc += p.bias[co];
    if (p.out_act == EVT_ACT_LRELU) acc = lrelu_f(acc, p.out_slope);
    else if (p.out_act == EVT_ACT_TANH) acc = tanhf(acc);
    if (p.res) acc += to_f<T>(reinterpret_cast<const T*>(p.res)[i]);
    reinterpret_cast<T*>(p.y)[i] = from_f<T>(acc);
  }
}

// backward-data: one thread per dx element (seq, row, ci); here p.x = dy, p.y_in = saved y, p.gate = saved x,
// p.res = dx_add, p.y = dx.
template <typename T>
__global__ void conv_naive_bwd_data(NvP p) {
  const long total = (long)p.nseq * p.lin * p.cin;
  const T* dy = reinterpret_cast<const T*>(p.x);
  const T* ys = reinterpret_cast<const T*>(p.y_in);
  const T* w = reinterpret_cast<const T*>(p.w);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ci = (int)(i % p.cin);
    const long sr = i / p.cin;
    const int row = (int)(sr % p.lin);
    const int seq = (int)(sr / p.lin);
    float acc = 0.f;
    if (!p.transposed) {
      const int cpg = p.cin / p.groups, opg = p.cout / p.groups;
      const int grp = ci / cpg, cl = ci - grp * cpg;
      for (int t = 0; t < p.k; ++t) {
        const int num = row + p.pad - t * p.dil;
        if (num < 0 || num % p.stride) continue;
        const int q = num / p.stride;
        if (q >= p.lout) continue;
        const long base = ((long)seq * p.lout + q) * p.cout + grp * opg;
        for (int o = 0; o < opg; ++o) {
          float d = to_f<T>(dy[base + o]);
          if (ys) d *= dact_from_out(p.out_act, to_f<T>(ys[base + o]), p.out_slope);
          acc += d * to_f<T>(w[reg_index(p, grp * opg + o, cl, t)]);
        }
      }
    } else {
      // dx[i][ci] = sum_{t,co} dy[i*s + t - pad][co] * W[ci][co][t]
      for (int t = 0; t < p.k; ++t) {
        const int orow = row * p.stride + t * p.dil - p.pad;
        if (orow < 0 || orow >= p.lout) continue;
        const long base = ((long)seq * p.lout + orow) * p.cout;
        for (int o = 0; o < p.cout; ++o) {
          float d = to_f<T>(dy[base + o]);
          if (ys) d *= dact_from_out(p.out_act, to_f<T>(ys[base + o]), p.out_slope);
          acc += d * to_f<T>(w[reg_index(p, ci, o, t)]);
        }
      }
    }
    if (p.gate) acc *= (to_f<T>(reinterpret_cast<const T*>(p.gate)[i]) > 0.f ? 1.f : p.in_slope);
    if (p.res) acc += to_f<T>(reinterpret_cast<const T*>(p.res)[i]);
    reinterpret_cast<T*>(p.y)[i] = from_f<T>(acc);
  }
}

// backward-weight: one block per REG weight element (d0, d1, t), positions strided over threads and
// blockIdx.y splits; p.x = saved x, p.res = dy, p.y_in = saved y.
template <typename T>
__global__ __launch_bounds__(256) void conv_naive_bwd_weight(NvP p) {
  __shared__ float red[4];
  const T* x = reinterpret_cast<const T*>(p.x);
  const T* dy = reinterpret_cast<const T*>(p.res);
  const T* ys = reinterpret_cast<const T*>(p.y_in);
  int e = blockIdx.x;
  const int t = e % p.k; e /= p.k;
  const int d1n = p.transposed ? p.cout : p.cin / p.groups;
  const int d1 = e % d1n;
  const int d0 = e / d1n;
  int co, ci;
  if (!p.transposed) {
    const int cpg = p.cin / p.groups, opg = p.cout / p.groups;
    co = d0; ci = (co / opg) * cpg + d1;
  } else { ci = d0; co = d1; }
  // positions enumerated on the side that is NOT strided: conv -> output rows q, convT -> input rows i
  const int nq = p.transposed ? p.lin : p.lout;
  const long total = (long)p.nseq * nq;
  float acc = 0.f;
  for (long i = (long)blockIdx.y * 256 + threadIdx.x; i < total; i += (long)p.nsplit * 256) {
    const int q = (int)(i % nq);
    const int seq = (int)(i / nq);
    const int other = q * p.stride + t * p.dil - p.pad;
    int xr, yr;
    if (!p.transposed) { xr = other; yr = q; if (xr < 0 || xr >= p.lin) continue; }
    else { xr = q; yr = other; if (yr < 0 || yr >= p.lout) continue; }
    const long yo = ((long)seq * p.lout + yr) * p.cout + co;
    float d = to_f<T>(dy[yo]);
    if (ys) d *= dact_from_out(p.out_act, to_f<T>(ys[yo]), p.out_slope);
    acc += d * lrelu_f(to_f<T>(x[((long)seq * p.lin + xr) * p.cin + ci]), p.in_slope);
  }
  acc = block_reduce_sum_256(acc, red);
  if (threadIdx.x == 0) atomicAdd(p.dw + reg_index(p, d0, d1, t), acc);
}

// dbias[c] += sum over rows of dy * act'(y); [rows][C] channels-last
template <typename T>
__global__ __launch_bounds__(256) void colsum_act(const T* dy, const T* ys, float* out, long rows, int C, int kind,
                                                  float slope, int rows_per_block, float* ws) {
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = min(rows, r0 + rows_per_block);
  for (int c = threadIdx.x; c < C; c += 256) {
    float acc = 0.f;
    for (long r = r0; r < r1; ++r) {
      float d = to_f<T>(dy[r * C + c]);
      if (ys) d *= dact_from_out(kind, to_f<T>(ys[r * C + c]), slope);
      acc += d;
    }
    if (ws) ws[(long)blockIdx.x * C + c] = acc;
    else atomicAdd(out + c, acc);
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

inline void pick_ck(int dtype, int b, int k, int stride_unused, int* ck, int* nchunk, int* kp) {
  (void)stride_unused;
  if (b % 32 == 0) { *ck = 32; *nchunk = b / 32; *kp = k; }
  else if (b % 16 == 0) { *ck = 16; *nchunk = b / 16; *kp = (dtype == EVT_DT_HALF) ? ((k + 1) & ~1) : k; }
  else { *ck = b; *nchunk = 1; *kp = k; }
}

template <typename T, int CK, int MT, int NT>
int launch_igemm_inst(const ConvP& p, int nphase, bool split, hipStream_t st) {
  constexpr int SZ = sizeof(T);
  constexpr int XROW = SZ == 4 ? CK * SZ + 16 : (CK == 16 ? 32 : 96);
  constexpr int WK = (SZ == 2 ? 256 : 128);
  constexpr int WROW = WK * SZ + (SZ == 4 ? 16 : 32);
  const int nreg = split ? NT : 1;
  const int Rr = ((split ? 16 : 16 * NT) - 1) * p.s_in + (p.KHp - 1) * p.dil + 1;
  const size_t lds = (size_t)16 * MT * WROW + (size_t)4 * nreg * Rr * XROW;
  if (lds > 160 * 1024) return EVT_ENOTSUP;
  // register prefetch pays only when there are several (chunk, tap-group) stages to overlap
  const int nst = p.nchunk * ceil_div(p.KHp, WK / CK);
  const bool pf = nst >= 3 || split;
  const void* fn;
  if constexpr (NT > 1) {
    fn = split ? reinterpret_cast<const void*>(&conv_igemm<T, CK, MT, NT, true, true>)
               : (pf ? reinterpret_cast<const void*>(&conv_igemm<T, CK, MT, NT, true, false>)
                     : reinterpret_cast<const void*>(&conv_igemm<T, CK, MT, NT, false, false>));
  } else {
    fn = pf ? reinterpret_cast<const void*>(&conv_igemm<T, CK, MT, NT, true, false>)
            : reinterpret_cast<const void*>(&conv_igemm<T, CK, MT, NT, false, false>);
  }
  static size_t max_set[3] = {0, 0, 0};
  const int vi = split ? 2 : (pf ? 1 : 0);
  if (lds > 48 * 1024 && lds > max_set[vi]) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return EVT_ELAUNCH;
    max_set[vi] = 160 * 1024;
  }
  const int gx = 8 * ceil_div(p.P, 8) * p.Y;
  evt_set_last_tag("conv_igemm<%s, %d, %d, %d, %s>", SZ == 2 ? "bf16" : "f32", CK, MT, NT,
                   split ? "split" : (pf ? "pf" : "sync"));
  if constexpr (NT > 1) {
    if (split) { hipLaunchKernelGGL((conv_igemm<T, CK, MT, NT, true, true>), dim3(gx, nphase), dim3(256), lds, st, p); return evt_check_launch(); }
  }
  if (pf) hipLaunchKernelGGL((conv_igemm<T, CK, MT, NT, true, false>), dim3(gx, nphase), dim3(256), lds, st, p);
  else hipLaunchKernelGGL((conv_igemm<T, CK, MT, NT, false, false>), dim3(gx, nphase), dim3(256), lds, st, p);
  return evt_check_launch();
}

template <typename T, int CK, int MT>
int launch_igemm_nt(const ConvP& p, int NT, int nphase, bool split, hipStream_t st) {
  switch (NT) {
    case 1: return launch_igemm_inst<T, CK, MT, 1>(p, nphase, false, st);
    case 2: return launch_igemm_inst<T, CK, MT, 2>(p, nphase, split, st);
    default: return launch_igemm_inst<T, CK, MT, 4>(p, nphase, split, st);
  }
}

template <typename T, int CK>
int launch_igemm_mt(const ConvP& p, int MT, int NT, int nphase, bool split, hipStream_t st) {
  switch (MT) {
    case 1: return launch_igemm_nt<T, CK, 1>(p, NT, nphase, split, st);
    case 2: return launch_igemm_nt<T, CK, 2>(p, NT, nphase, split, st);
    default: return launch_igemm_nt<T, CK, 4>(p, NT, nphase, split, st);
  }
}

// Generic igemm launch.  A = output channels, B = K-side channels.
int launch_igemm(int dtype, ConvP p, int A, int B, int nphase, hipStream_t st, bool allow_deep) {
  if (A % 16 || B % 16) return EVT_ENOTSUP;
  // specialised paths first; EVT_ENOTSUP from one of them means "not this shape after all": fall through to the generic kernel
  if (allow_deep && evt_conv::deep_eligible(p, dtype, A, B, nphase)) {
    const int rc = evt_conv::launch_conv_deep(p, A, B, nphase, st);
    if (rc != EVT_ENOTSUP) return rc;
  }
  if (allow_deep && evt_conv::narrow_eligible(p, dtype, A, B, nphase)) {
    const int rc = evt_conv::launch_conv_narrow(p, A, B, nphase, st);
    if (rc != EVT_ENOTSUP) return rc;
  }
  const int CK = (B % 32 == 0) ? 32 : 16;
  const int MT = (A % 64 == 0) ? 4 : (A % 32 == 0 ? 2 : 1);
  p.Y = A / (16 * MT);
  // Short sequences (a 64-position unit would be mostly padding): 16-position units from any sequence, NT per wave.
  // Long sequences: the largest contiguous unit that still gives the chip ~2 blocks per CU.
  const bool split = p.Q < 56 && dtype == EVT_DT_HALF;
  int NT = 4;
  if (split) {
    const long units16 = (long)p.nseq * ceil_div(p.Q, 16);
    while (NT > 1 && ((units16 + 4 * NT - 1) / (4 * NT)) * p.Y * nphase < 384) NT >>= 1;
  } else {
    while (NT > 1) {
      const long units = (long)p.nseq * ceil_div(p.Q, 16 * NT);
      const long blocks = ((units + 3) / 4) * p.Y * nphase;
      const int waste = ceil_div(p.Q, 16 * NT) * 16 * NT - p.Q;
      if (blocks >= 512 && waste * 3 <= p.Q) break;
      NT >>= 1;
    }
  }
  for (;; NT >>= 1) {
    const bool sp = split && NT > 1;
    p.U = ceil_div(p.Q, sp ? 16 : 16 * NT);
    const long units = (long)p.nseq * p.U;
    p.P = (int)((units + (sp ? 4 * NT : 4) - 1) / (sp ? 4 * NT : 4));
    int rc;
    if (dtype == EVT_DT_HALF)
      rc = (CK == 32) ? launch_igemm_mt<h16_t, 32>(p, MT, NT, nphase, sp, st) : launch_igemm_mt<h16_t, 16>(p, MT, NT, nphase, sp, st);
    else
      rc = (CK == 32) ? launch_igemm_mt<float, 32>(p, MT, NT, nphase, sp, st) : launch_igemm_mt<float, 16>(p, MT, NT, nphase, sp, st);
    if (rc != EVT_ENOTSUP || NT == 1) return rc;  // ENOTSUP here = LDS too large: shrink the unit
  }
}

template <typename T, int CK>
int launch_wgrad_inst(const WgP& p, hipStream_t st) {
  const int BR = 63 * p.s + 3 * p.dil + 1;
  const size_t lds = (size_t)(64 * 66 + (size_t)BR * (CK + 2)) * sizeof(T);
  if (lds > 160 * 1024) return EVT_ENOTSUP;
  static bool attr = false;
  if (lds > 48 * 1024 && !attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad<T, CK>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return EVT_ELAUNCH;
    attr = true;
  }
  const int gx = ceil_div(p.CA, 64) * p.nchunk * p.ntapgrp;
  evt_set_last_tag("conv_wgrad<%s, %d>", sizeof(T) == 2 ? "bf16" : "f32", CK);
  hipLaunchKernelGGL((conv_wgrad<T, CK>), dim3(gx, p.nsplit), dim3(256), lds, st, p);
  return evt_check_launch();
}

template <int CK, int TA>
int launch_wgrad_tr_inst(const WgP& p, hipStream_t st) {
  constexpr int PK = 64 * (4 / (TA / 16));
  const int BR = (PK - 1) * p.s + 3 * p.dil + 1;
  size_t lds = ((size_t)PK * (TA + 8) + (size_t)BR * (CK + 8)) * 2;
  if (TA < 64 && lds < (size_t)16384 * (CK / 16)) lds = (size_t)16384 * (CK / 16);  // cross-wave reduction scratch
  if (lds > 150 * 1024) return EVT_ENOTSUP;
  static bool attr = false;
  if (lds > 48 * 1024 && !attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_tr<CK, TA>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return EVT_ELAUNCH;
    attr = true;
  }
  const int gx = (p.CA / TA) * p.nchunk * p.ntapgrp;
  evt_set_last_tag("conv_wgrad_tr<%d, %d>", CK, TA);
  hipLaunchKernelGGL((conv_wgrad_tr<CK, TA>), dim3(gx, p.nsplit), dim3(256), lds, st, p);
  int rc = evt_check_launch();
  if (rc || !p.ws || p.parts > 0) return rc;
  const long img = (long)p.CA * p.nchunk * p.KHp * CK;
  rc = evt_conv::launch_fold_partials(p.ws, img, p.nsplit, p.dw, img, st);
  if (rc || !p.dbias) return rc;
  return evt_conv::launch_fold_partials(p.ws + (long)p.nsplit * img, p.CA, p.nsplit, p.dbias, p.CA, st);
}

// bf16 only; returns ENOTSUP when the tile does not fit so the caller can use the gather kernel
int launch_wgrad_tr(WgP p, hipStream_t st) {
  if (p.CB % 16 || p.CA % 16) return EVT_ENOTSUP;
  const int CK = (p.CB % 32 == 0) ? 32 : 16;
  const int TA = (p.CA % 64 == 0) ? 64 : (p.CA % 32 == 0 ? 32 : 16);
  const int PK = 64 * (4 / (TA / 16));
  p.nchunk = p.CB / CK;
  p.ntapgrp = ceil_div(p.KHp, 4);
  const long iters = (long)p.nseq * ceil_div(p.Q, PK);
  const long tiles = (long)(p.CA / TA) * p.nchunk * p.ntapgrp;
  // enough blocks to fill the chip, but a bounded number of atomic partials per dW element
  long split = (768 + tiles - 1) / tiles;
  if (split > 256) split = 256;
  if (split > iters) split = iters;
  if (split < 1) split = 1;
  // slab mode only when the caller's slabs cover the split this kernel wants (it is latency-bound: fewer, longer blocks
  // cost more than the second launch of the scratch-row mode saves -- measured: 28 -> 64 us on the 32 -> 128 k5 s3 layer
  // with 32 slabs instead of 192 splits)
  if (p.parts > 0 && split > p.parts) p.parts = 0;
  if (p.parts > 0) {
    // one slab per split (the caller sizes them by image: up to 256 for the 11 - 45 KB images of the narrow vocoder
    // stages); no second launch
    p.nsplit = (int)split;
    p.now_used = p.prev_used > p.nsplit ? p.prev_used : p.nsplit;
    if (p.used_host) *p.used_host = p.now_used;
    if (!p.dbias) p.db_part = nullptr;
  } else if (p.ws) {
    // every split stores a partial image (+ bias row); rows of one block's tile that it does not own stay unwritten,
    // so all (tile, split) blocks must exist: split <= iters holds above
    const long row = (long)p.CA * p.nchunk * p.KHp * CK + p.CA;
    if (row * split > p.ws_floats) split = p.ws_floats / row;
    if (split < 2) p.ws = nullptr;                 // nothing to fold: accumulate directly
    if (split < 1) split = 1;
  }
  p.nsplit = (int)split;
  if (CK == 32) {
    if (TA == 64) return launch_wgrad_tr_inst<32, 64>(p, st);
    if (TA == 32) return launch_wgrad_tr_inst<32, 32>(p, st);
    return launch_wgrad_tr_inst<32, 16>(p, st);
  }
  if (TA == 64) return launch_wgrad_tr_inst<16, 64>(p, st);
  if (TA == 32) return launch_wgrad_tr_inst<16, 32>(p, st);
  return launch_wgrad_tr_inst<16, 16>(p, st);
}

int launch_wgrad(int dtype, WgP p, hipStream_t st) {
  if (p.CB % 16) return EVT_ENOTSUP;
  const int CK = (p.CB % 32 == 0) ? 32 : 16;
  p.nchunk = p.CB / CK;
  p.ntapgrp = ceil_div(p.KHp, 4);
  const long iters = (long)p.nseq * ceil_div(p.Q, 64);
  const long tiles = (long)ceil_div(p.CA, 64) * p.nchunk * p.ntapgrp;
  long split = (2048 + tiles - 1) / tiles;  // aim for >= ~2048 blocks
  if (split > iters) split = iters;
  if (split < 1) split = 1;
  if (split > 65535) split = 65535;
  p.nsplit = (int)split;
  if (dtype == EVT_DT_HALF) return CK == 32 ? launch_wgrad_inst<h16_t, 32>(p, st) : launch_wgrad_inst<h16_t, 16>(p, st);
  return CK == 32 ? launch_wgrad_inst<float, 32>(p, st) : launch_wgrad_inst<float, 16>(p, st);
}

NvP make_nvp(const evt_conv1d_params* c) {
  NvP p{};
  p.nseq = c->nseq; p.lin = c->lin; p.lout = evt_conv1d_lout(c); p.cin = c->cin; p.cout = c->cout;
  p.k = c->k; p.stride = c->stride; p.pad = c->pad; p.dil = c->dil; p.groups = c->groups;
  p.transposed = c->transposed; p.in_slope = c->in_slope; p.out_act = c->out_act; p.out_slope = c->out_slope;
  evt_wlayout l; evt_conv1d_layout(c, &l);
  p.ck = l.reg_ck; p.nchunk = l.reg_nchunk; p.kp = l.reg_kp;
  p.nsplit = 1;
  return p;
}

bool igemm_ok(const evt_conv1d_params* c) {
  if (c->groups != 1) return false;
  if (c->cin % 16 || c->cout % 16) return false;
  if (c->stride > 1 && c->dil != 1) return false;
  return true;
}

int valid(const evt_conv1d_params* c) {
  if (!c || c->nseq <= 0 || c->lin <= 0 || c->cin <= 0 || c->cout <= 0 || c->k <= 0 || c->stride <= 0 ||
      c->dil <= 0 || c->groups <= 0 || c->pad < 0)
    return EVT_EINVAL;
  if (c->dtype != EVT_DT_F32 && c->dtype != EVT_DT_HALF) return EVT_EINVAL;
  if (c->cin % c->groups || c->cout % c->groups) return EVT_EINVAL;
  if (c->transposed && (c->groups != 1 || c->dil != 1)) return EVT_ENOTSUP;
  if (evt_conv1d_lout(c) <= 0) return EVT_EINVAL;
  return EVT_OK;
}

}  // namespace

extern "C" {

int32_t evt_conv1d_lout(const evt_conv1d_params* c) {
  if (!c->transposed) return (c->lin + 2 * c->pad - c->dil * (c->k - 1) - 1) / c->stride + 1;
  return (c->lin - 1) * c->stride - 2 * c->pad + c->dil * (c->k - 1) + 1;
}

int32_t evt_conv1d_wants_plain_dy(const evt_conv1d_params* c) {
  if (!c || valid(c) != EVT_OK) return 0;
  if (c->impl != EVT_IMPL_AUTO || c->dtype != EVT_DT_HALF || c->transposed || !igemm_ok(c)) return 0;
  if (c->out_act == EVT_ACT_NONE || c->in_slope != 1.f) return 0;
  // same descriptor geometry evt_conv1d_bwd_data builds (K side = cout, output channels = cin)
  ConvP p{};
  p.in_slope = 1.f;
  p.nseq = c->nseq;
  p.nchunk = c->cout / 32;
  int nphase = 1;
  if (c->stride == 1) p.Q = c->lin;
  else { nphase = c->stride; p.Q = (c->lin - 1 + c->pad) / c->stride + 1; }
  if (evt_conv::deep_eligible(p, c->dtype, c->cin, c->cout, nphase)) return 1;
  {  // the stride-1 narrow kernel takes plain operands too
    evt_wlayout l; evt_conv1d_layout(c, &l);
    ConvP q = p;
    q.s_in = q.s_out = 1; q.off_out = 0; q.KHp = l.alt_kp; q.nchunk = l.alt_nchunk;
    if (c->stride == 1 && evt_conv::narrow_eligible(q, c->dtype, c->cin, c->cout, 1)) return 1;
  }
  // ... or the weight gradient does (A = dy [nseq][lout][cout], B = x)
  WgP w{};
  w.nseq = c->nseq; w.KH = w.KHp = c->k; w.Q = w.LA = evt_conv1d_lout(c); w.CA = c->cout; w.CB = c->cin;
  w.s = c->stride; w.dil = c->dil; w.LB = c->lin;
  w.a_slope = w.b_slope = 1.f;
  return (evt_conv::wgrad_deep_eligible(w, c->dtype) || evt_conv::wgrad_halo_eligible(w, c->dtype)) ? 1 : 0;
}

int32_t evt_conv1d_wants_plain_x(const evt_conv1d_params* c) {
  if (!c || valid(c) != EVT_OK) return 0;
  if (c->impl != EVT_IMPL_AUTO || c->dtype != EVT_DT_HALF || c->in_slope == 1.f || !igemm_ok(c)) return 0;
  if (evt_grouped_supported(c) || evt_small_kind(c) != 0) return 0;
  // same descriptor geometry evt_conv1d_fwd builds, with the load-side activation taken out
  evt_wlayout l; evt_conv1d_layout(c, &l);
  const int lout = evt_conv1d_lout(c);
  ConvP p{};
  p.in_slope = 1.f;
  p.nseq = c->nseq;
  int nphase = 1;
  if (!c->transposed) { p.nchunk = l.reg_nchunk; p.KHp = l.reg_kp; p.Q = lout; }
  else {
    p.nchunk = l.alt_nchunk; p.KHp = l.alt_kp;
    if (c->stride == 1) p.Q = lout;
    else { nphase = c->stride; p.Q = (lout - 1 + c->pad) / c->stride + 1; }
  }
  return evt_conv::deep_eligible(p, c->dtype, c->cout, c->cin, nphase) ? 1 : 0;
}

int evt_conv1d_layout(const evt_conv1d_params* c, evt_wlayout* o) {
  if (!c || !o) return EVT_EINVAL;
  const int d0 = c->transposed ? c->cin : c->cout;
  const int d1 = c->transposed ? c->cout : c->cin / c->groups;
  o->d0 = d0; o->d1 = d1; o->k = c->k; o->stride = c->stride;
  pick_ck(c->dtype, d1, c->k, 1, &o->reg_ck, &o->reg_nchunk, &o->reg_kp);
  o->reg_elems = (int64_t)d0 * o->reg_nchunk * o->reg_kp * o->reg_ck;
  if (c->stride == 1) {
    pick_ck(c->dtype, d0, c->k, 1, &o->alt_ck, &o->alt_nchunk, &o->alt_kp);
    o->alt_nphase = 1;
  } else {
    const int J = (c->k + c->stride - 1) / c->stride;
    pick_ck(c->dtype, d0, J, 1, &o->alt_ck, &o->alt_nchunk, &o->alt_kp);
    o->alt_nphase = c->stride;
  }
  o->alt_elems = (int64_t)o->alt_nphase * d1 * o->alt_nchunk * o->alt_kp * o->alt_ck;
  return EVT_OK;
}

int evt_conv1d_fwd(const evt_conv1d_params* c, const void* x, const void* w_reg, const void* w_alt,
                   const float* bias, const void* res, void* y, void* stream) {
  int rc = valid(c);
  if (rc) return rc;
  if (!x || !y) return EVT_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const bool use_igemm = c->impl != EVT_IMPL_NAIVE && igemm_ok(c);
  if (c->impl == EVT_IMPL_IGEMM && !use_igemm) return EVT_ENOTSUP;
  const int lout = evt_conv1d_lout(c);
  evt_wlayout l; evt_conv1d_layout(c, &l);
  // a handful of rows (a 1x1 layer applied to one vector per item): weight rows on the MFMA's M side (rows_gemm.hip)
  if (w_reg && evt_conv::rows16_eligible(c, c->nseq * c->lin, c->cout, c->cin,
                                         res || c->in_slope != 1.f || c->out_act != EVT_ACT_NONE))
    return evt_conv::launch_rows16(x, w_reg, bias, y, c->nseq * c->lin, c->cout, c->cin, st);
  evt_set_last_tag("conv_naive_fwd");
  if (c->impl != EVT_IMPL_NAIVE && !res && evt_grouped_supported(c)) {
    if (!w_reg) return EVT_EINVAL;
    evt_set_last_tag("grouped_fwd");
    return evt_grouped_fwd(c, x, w_reg, bias, y, stream);
  }
  if (c->impl != EVT_IMPL_NAIVE && !res && evt_small_kind(c) == 1) {
    if (!w_reg) return EVT_EINVAL;
    return evt_cout1_fwd(c, x, w_reg, bias, y, stream);
  }
  if (c->impl != EVT_IMPL_NAIVE && !res && evt_small_kind(c) == 2) {
    if (!w_reg) return EVT_EINVAL;
    return evt_cin1_fwd(c, x, w_reg, bias, y, stream);
  }
  if (!use_igemm) {
    if (!w_reg) return EVT_EINVAL;
    NvP p = make_nvp(c);
    p.x = x; p.w = w_reg; p.bias = bias; p.res = res; p.y = y;
    const long total = (long)c->nseq * lout * c->cout;
    const int blocks = (int)((total + 255) / 256 > 65535 * 8 ? 65535 * 8 : (total + 255) / 256);
    if (c->dtype == EVT_DT_HALF) hipLaunchKernelGGL(conv_naive_fwd<h16_t>, dim3(blocks), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(conv_naive_fwd<float>, dim3(blocks), dim3(256), 0, st, p);
    return evt_check_launch();
  }
  const void* w = c->transposed ? w_alt : w_reg;
  if (!w) return EVT_EINVAL;
  ConvP p{};
  p.x = x; p.xact = nullptr; p.w = w; p.bias = bias; p.res = res; p.gate = nullptr; p.y = y;
  p.nseq = c->nseq; p.Lin = c->lin; p.Lout = lout; p.Cin = c->cin; p.Cout = c->cout;
  p.in_slope = c->in_slope; p.xact_kind = 0; p.xact_slope = 1.f; p.out_act = c->out_act; p.out_slope = c->out_slope;
  p.gate_slope = 1.f;
  int nphase = 1;
  if (!c->transposed) {
    p.KHp = l.reg_kp; p.nchunk = l.reg_nchunk;
    p.s_in = c->stride; p.dil = c->dil; p.off_in = -c->pad;
    p.s_out = 1; p.off_out = 0; p.off_out_phase = 0; p.Q = lout; p.w_phase_stride = 0;
  } else {
    p.KHp = l.alt_kp; p.nchunk = l.alt_nchunk; p.dil = 1; p.s_in = 1;
    if (c->stride == 1) {
      p.off_in = c->pad - (c->k - 1); p.s_out = 1; p.off_out = 0; p.off_out_phase = 0; p.Q = lout;
      p.w_phase_stride = 0;
    } else {
      const int J = (c->k + c->stride - 1) / c->stride;
      nphase = c->stride;
      p.off_in = -(J - 1); p.s_out = c->stride; p.off_out = -c->pad; p.off_out_phase = 1;
      p.Q = (lout - 1 + c->pad) / c->stride + 1;
      p.w_phase_stride = (long)c->cout * l.alt_nchunk * l.alt_kp * l.alt_ck;
    }
  }
  return launch_igemm(c->dtype, p, c->cout, c->cin, nphase, st, c->impl == EVT_IMPL_AUTO);
}

int evt_conv1d_bwd_data(const evt_conv1d_params* c, const void* dy, const void* y, const void* w_reg,
                        const void* w_alt, const void* x, const void* dx_add, void* dx, void* stream) {
  int rc = valid(c);
  if (rc) return rc;
  if (!dy || !dx) return EVT_EINVAL;
  if (c->out_act != EVT_ACT_NONE && !y) return EVT_EINVAL;
  if (c->in_slope != 1.f && !x) return EVT_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const bool use_igemm = c->impl != EVT_IMPL_NAIVE && igemm_ok(c);
  if (c->impl == EVT_IMPL_IGEMM && !use_igemm) return EVT_ENOTSUP;
  const int lout = evt_conv1d_lout(c);
  evt_wlayout l; evt_conv1d_layout(c, &l);
  const void* ysv = c->out_act != EVT_ACT_NONE ? y : nullptr;
  const void* gate = c->in_slope != 1.f ? x : nullptr;
  if (w_alt && evt_conv::rows16_eligible(c, c->nseq * c->lin, c->cin, c->cout, ysv || gate || dx_add))
    return evt_conv::launch_rows16(dy, w_alt, nullptr, dx, c->nseq * c->lin, c->cin, c->cout, st);
  evt_set_last_tag("conv_naive_bwd_data");
  if (c->impl != EVT_IMPL_NAIVE && !gate && !dx_add && evt_grouped_supported(c)) {
    if (!w_reg) return EVT_EINVAL;
    evt_set_last_tag("grouped_bwd_data");
    return evt_grouped_bwd_data(c, dy, y, w_reg, dx, stream);
  }
  if (c->impl != EVT_IMPL_NAIVE && evt_small_kind(c) != 0) {
    if (!w_reg) return EVT_EINVAL;
    rc = evt_small_kind(c) == 1 ? evt_cout1_bwd_data(c, dy, y, w_reg, gate, dx_add, dx, stream)
                                : evt_cin1_bwd_data(c, dy, y, w_reg, gate, dx_add, dx, stream);
    if (rc != EVT_ENOTSUP) return rc;
  }
  if (!use_igemm) {
    if (!w_reg) return EVT_EINVAL;
    NvP p = make_nvp(c);
    p.x = dy; p.y_in = ysv; p.w = w_reg; p.gate = gate; p.res = dx_add; p.y = dx;
    const long total = (long)c->nseq * c->lin * c->cin;
    const int blocks = (int)((total + 255) / 256 > 65535 * 8 ? 65535 * 8 : (total + 255) / 256);
    if (c->dtype == EVT_DT_HALF) hipLaunchKernelGGL(conv_naive_bwd_data<h16_t>, dim3(blocks), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(conv_naive_bwd_data<float>, dim3(blocks), dim3(256), 0, st, p);
    return evt_check_launch();
  }
  const void* w = c->transposed ? w_reg : w_alt;
  if (!w) return EVT_EINVAL;
  ConvP p{};
  p.x = dy; p.xact = ysv; p.w = w; p.bias = nullptr; p.res = dx_add; p.gate = gate; p.y = dx;
  p.nseq = c->nseq; p.Lin = lout; p.Lout = c->lin; p.Cin = c->cout; p.Cout = c->cin;
  p.in_slope = 1.f; p.xact_kind = c->out_act; p.xact_slope = c->out_slope; p.out_act = EVT_ACT_NONE;
  p.out_slope = 1.f; p.gate_slope = c->in_slope;
  int nphase = 1;
  if (!c->transposed) {
    p.KHp = l.alt_kp; p.nchunk = l.alt_nchunk;
    if (c->stride == 1) {
      p.s_in = 1; p.dil = c->dil; p.off_in = c->pad - (c->k - 1) * c->dil;
      p.s_out = 1; p.off_out = 0; p.off_out_phase = 0; p.Q = c->lin; p.w_phase_stride = 0;
    } else {
      const int J = (c->k + c->stride - 1) / c->stride;
      nphase = c->stride;
      p.s_in = 1; p.dil = 1; p.off_in = -(J - 1);
      p.s_out = c->stride; p.off_out = -c->pad; p.off_out_phase = 1;
      p.Q = (c->lin - 1 + c->pad) / c->stride + 1;
      p.w_phase_stride = (long)c->cin * l.alt_nchunk * l.alt_kp * l.alt_ck;
    }
  } else {
    // dx[i][ci] = sum_t sum_co dy[i*s + t - pad][co] W[ci][co][t]  (REG image, rows = ci)
    p.KHp = l.reg_kp; p.nchunk = l.reg_nchunk;
    p.s_in = c->stride; p.dil = 1; p.off_in = -c->pad;
    p.s_out = 1; p.off_out = 0; p.off_out_phase = 0; p.Q = c->lin; p.w_phase_stride = 0;
  }
  return launch_igemm(c->dtype, p, c->cin, c->cout, nphase, st, c->impl == EVT_IMPL_AUTO);
}

int evt_conv1d_bwd_weight(const evt_conv1d_params* c, const void* x, const void* dy, const void* y, float* dw,
                          float* dbias, void* stream) {
  return evt_conv1d_bwd_weight_parts(c, x, dy, y, dw, dbias, nullptr, stream);
}

int evt_conv1d_bwd_weight_parts(const evt_conv1d_params* c, const void* x, const void* dy, const void* y, float* dw,
                                float* dbias, evt_wgrad_parts* sp, void* stream) {
  int rc = valid(c);
  if (rc) return rc;
  if (!x || !dy || !dw) return EVT_EINVAL;
  int used_host = 0;
  if (sp) {
    sp->used = 0;
    if (sp->parts < 1 || !sp->used_dev || sp->prev_used < 0 || sp->prev_used > sp->parts) return EVT_EINVAL;
    if (sp->parts > 1 && (!sp->dw_extra || sp->part_stride <= 0)) return EVT_EINVAL;
    if (sp->ws && sp->ws_floats <= 0) return EVT_EINVAL;
  }
  if (c->out_act != EVT_ACT_NONE && !y) return EVT_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int lout = evt_conv1d_lout(c);
  const void* ysv = c->out_act != EVT_ACT_NONE ? y : nullptr;
  evt_wlayout l; evt_conv1d_layout(c, &l);
  const bool grouped = c->impl != EVT_IMPL_NAIVE && evt_grouped_supported(c);
  const bool igemm_path = !grouped && c->impl != EVT_IMPL_NAIVE && igemm_ok(c) && l.reg_kp <= 64;
  // descriptor of the MFMA weight-gradient kernels
  WgP p{};
  p.dw = dw; p.nseq = c->nseq; p.KH = c->k; p.KHp = l.reg_kp; p.s = c->stride; p.dil = c->dil; p.off = -c->pad;
  if (!c->transposed) {
    // dW[co][.][t][ci] += dy_eff[q][co] * lrelu(x)[q*s + t*dil - pad][ci]
    p.A = dy; p.Aact = ysv; p.aact_kind = c->out_act; p.aact_slope = c->out_slope; p.a_slope = 1.f;
    p.B = x; p.Bact = nullptr; p.bact_kind = 0; p.bact_slope = 1.f; p.b_slope = c->in_slope;
    p.LA = lout; p.CA = c->cout; p.LB = c->lin; p.CB = c->cin; p.Q = lout;
  } else {
    // dW[ci][.][t][co] += lrelu(x)[i][ci] * dy_eff[i*s + t - pad][co]
    p.A = x; p.Aact = nullptr; p.aact_kind = 0; p.aact_slope = 1.f; p.a_slope = c->in_slope;
    p.B = dy; p.Bact = ysv; p.bact_kind = c->out_act; p.bact_slope = c->out_slope; p.b_slope = 1.f;
    p.LA = c->lin; p.CA = c->cin; p.LB = lout; p.CB = c->cout; p.Q = c->lin;
  }
  p.dbias = nullptr;
  if (sp) {
    p.dw_extra = sp->dw_extra; p.part_stride = sp->part_stride; p.parts = sp->parts; p.prev_used = sp->prev_used;
    p.db_part = sp->db_part; p.used = sp->used_dev; p.used_host = &used_host; p.dirty0 = sp->dirty0 || sp->prev_used > 0;
  }
  // wide layers with plain operands: LDS-DMA GEMM kernel (conv_deep.hip); its dbias comes from the column-sum kernel
  // dense layers (k = 1, long reductions): the 128 x 128 weight-gradient GEMM tile
  const bool gemm_w = igemm_path && c->impl == EVT_IMPL_AUTO && !c->transposed && evt_conv::wgrad_gemm_eligible(p, c->dtype);
  // stride-1 layers with 3..11 taps and long sequences: one staged window of x serves all taps (wgrad_halo.hip)
  const bool halo_w = !gemm_w && igemm_path && c->impl == EVT_IMPL_AUTO && evt_conv::wgrad_halo_eligible(p, c->dtype);
  const bool deep_w = !gemm_w && !halo_w && igemm_path && c->impl == EVT_IMPL_AUTO && evt_conv::wgrad_deep_eligible(p, c->dtype);
  // latency-bound mid-size layers: ring-pipelined LDS-DMA kernel (fuses dbias when A is dy)
  const bool ring_w = !deep_w && !halo_w && igemm_path && c->impl == EVT_IMPL_AUTO && evt_conv::wgrad_ring_eligible(p, c->dtype);
  // dbias is fused into the bf16 MFMA weight-gradient kernel when its A operand is dy (plain Conv1d)
  const bool cin1 = !grouped && c->impl != EVT_IMPL_NAIVE && evt_small_kind(c) == 2;   // fuses dbias as well
  const bool fuse_bias = dbias && ((igemm_path && c->dtype == EVT_DT_HALF && !c->transposed) || cin1);
  float* ws = sp ? sp->ws : nullptr;
  const long ws_floats = sp ? (long)sp->ws_floats : 0;
  if (dbias && !fuse_bias) {
    const long rows = (long)c->nseq * lout;
    const int V = c->dtype == EVT_DT_HALF ? 8 : 4;
    int blocks;
    float* wsb = nullptr;
    if (c->cout % V == 0 && c->cout / V <= 256) {
      long rpb = (rows + 255) / 256;   // <= 256 blocks: every block ends with one partial sum per channel
      if (rpb < 16) rpb = 16;
      blocks = (int)((rows + rpb - 1) / rpb);
      if (ws && blocks >= 2 && (long)blocks * c->cout <= ws_floats) wsb = ws;
      if (c->dtype == EVT_DT_HALF)
        hipLaunchKernelGGL(colsum_act2<h16_t>, dim3(blocks), dim3(256), 0, st, (const h16_t*)dy, (const h16_t*)ysv,
                           dbias, rows, c->cout, c->out_act, c->out_slope, (int)rpb, wsb);
      else
        hipLaunchKernelGGL(colsum_act2<float>, dim3(blocks), dim3(256), 0, st, (const float*)dy, (const float*)ysv,
                           dbias, rows, c->cout, c->out_act, c->out_slope, (int)rpb, wsb);
    } else {
      const int rpb = 64;
      blocks = (int)((rows + rpb - 1) / rpb);
      if (ws && blocks >= 2 && (long)blocks * c->cout <= ws_floats) wsb = ws;
      if (c->dtype == EVT_DT_HALF)
        hipLaunchKernelGGL(colsum_act<h16_t>, dim3(blocks), dim3(256), 0, st, (const h16_t*)dy, (const h16_t*)ysv,
                           dbias, rows, c->cout, c->out_act, c->out_slope, rpb, wsb);
      else
        hipLaunchKernelGGL(colsum_act<float>, dim3(blocks), dim3(256), 0, st, (const float*)dy, (const float*)ysv,
                           dbias, rows, c->cout, c->out_act, c->out_slope, rpb, wsb);
    }
    rc = evt_check_launch();
    if (rc) return rc;
    if (wsb) {
      rc = evt_conv::launch_fold_partials(wsb, c->cout, blocks, dbias, c->cout, st);
      if (rc) return rc;
    }
  }
  evt_set_last_tag("conv_naive_bwd_weight");
  if (grouped) {
    evt_set_last_tag("grouped_bwd_weight");
    return evt_grouped_bwd_weight(c, x, dy, y, dw, stream);
  }
  if (c->impl != EVT_IMPL_NAIVE && evt_small_kind(c) == 1) return evt_cout1_bwd_weight(c, x, dy, y, dw, ws, ws_floats, stream);
  if (cin1) return evt_cin1_bwd_weight(c, x, dy, y, dw, dbias, ws, ws_floats, stream);
  const bool use_igemm = igemm_path;
  if (c->impl == EVT_IMPL_IGEMM && !use_igemm) return EVT_ENOTSUP;
  if (!use_igemm) {
    NvP p = make_nvp(c);
    p.x = x; p.res = dy; p.y_in = ysv; p.dw = dw;
    const int d1n = c->transposed ? c->cout : c->cin / c->groups;
    const int d0n = c->transposed ? c->cin : c->cout;
    const long elems = (long)d0n * d1n * c->k;
    const long pos = (long)c->nseq * (c->transposed ? c->lin : lout);
    long split = (4096 + elems - 1) / elems;
    const long maxsplit = (pos + 2047) / 2048;
    if (split > maxsplit) split = maxsplit;
    if (split < 1) split = 1;
    p.nsplit = (int)split;
    if (c->dtype == EVT_DT_HALF)
      hipLaunchKernelGGL(conv_naive_bwd_weight<h16_t>, dim3((unsigned)elems, p.nsplit), dim3(256), 0, st, p);
    else
      hipLaunchKernelGGL(conv_naive_bwd_weight<float>, dim3((unsigned)elems, p.nsplit), dim3(256), 0, st, p);
    return evt_check_launch();
  }
  if (gemm_w) {
    p.dbias = fuse_bias ? dbias : nullptr;
    p.parts = 0;                                     // no slab mode: the s1 dense layers accumulate into the arena
    return evt_conv::launch_wgrad_gemm(p, st);
  }
  if (halo_w || deep_w || ring_w) {
    p.dbias = fuse_bias ? dbias : nullptr;
    if (!p.dbias) p.db_part = nullptr;
    rc = halo_w ? evt_conv::launch_wgrad_halo(p, st) : (deep_w ? evt_conv::launch_wgrad_deep(p, st) : evt_conv::launch_wgrad_ring(p, st));
    if (sp) sp->used = used_host;
    return rc;
  }
  if (c->dtype == EVT_DT_HALF) {
    p.dbias = fuse_bias ? dbias : nullptr;
    p.ws = ws; p.ws_floats = ws_floats;
    if (p.parts < 2) p.parts = 0;                  // one slab = no split possible: scratch rows (or atomics) instead
    rc = launch_wgrad_tr(p, st);
    if (sp && rc != EVT_ENOTSUP) sp->used = used_host;
    if (rc != EVT_ENOTSUP) return rc;
    p.parts = 0;
    if (fuse_bias) {  // tile did not fit: the gather kernel has no fused bias
      const long rows = (long)c->nseq * lout;
      hipLaunchKernelGGL(colsum_act<h16_t>, dim3((int)((rows + 63) / 64)), dim3(256), 0, st, (const h16_t*)dy,
                         (const h16_t*)ysv, dbias, rows, c->cout, c->out_act, c->out_slope, 64, (float*)nullptr);
      rc = evt_check_launch();
      if (rc) return rc;
    }
  }
  p.dbias = nullptr;
  return launch_wgrad(c->dtype, p, st);
}

}  // extern "C"
