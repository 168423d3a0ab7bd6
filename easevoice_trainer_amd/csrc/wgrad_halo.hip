// wgrad_halo: weight gradient of the stride-1 layers with 3..11 taps (gfx950).
//
// Reference call sites: the backward of Conv1d in HiFi-GAN's ResBlock1 (k = 3 / 7 / 11, dilation 1 / 3 / 5,
// src/easevoice/module/modules.py:226-311), of WN's in_layers (k = 5, modules.py:154-212) and of the encoders' FFN
// (k = 3, attentions.py:383-420), as torch.autograd computes them for `F.conv1d`.
//
//   dW[a][chunk][t][cc] += sum over (seq, q) of dy[seq][q][a] * x[seq][q + t*dil + off][chunk*32 + cc]
//
// GEMM view: M = dy channels, N = (tap, x channel), K = positions.  wgrad_deep / wgrad_ring stage one [64 pos][32 ch]
// tile of x PER TAP: for k = 11 that is 11 copies of (nearly) the same rows through the LDS-DMA path, 73 flop per staged
// byte, and the kernels sit at 150-310 TFLOP/s on the vocoder stages, bound by latency x bytes in flight.  Here a K
// stage is 64 consecutive positions of ONE sequence, and the x operand is staged once as the window those positions see
// through all taps: rows q0 + off .. q0 + off + 63 + (k-1)*dil (<= 128 rows of 64 bytes).  Tap t of position k is window
// row k + t*dil: the taps are row displacements of the same LDS tile, i.e. per-lane addresses of the transpose reads.
// One block = (32*MA dy channels) x (ALL taps x 32 x channels): 230-275 flop per staged byte for k = 11.
//   LDS stage: A tile [64 pos][32*MA ch] (dy, swizzled as in wgrad_ring) + window [128 rows][32 ch];
//   ring of NS stages, counted vmcnt + one raw barrier per stage; fragments by ds_read_b64_tr_b16 (builtin: the tap
//   displacement is a runtime value, so the addresses are computed, not instruction offsets).
//   window swizzle (16-byte slots of a 64-byte row): slot ^= 2 * ((row >> 3) & 1), keyed on the WINDOW row, so it is the
//   same function for the DMA (which writes row j) and for a tap read (which reads row k + t*dil).
// Sequence tails (Q not a multiple of 64) are zero rows of A; rows of the window outside the sequence (conv padding) read
// the zero page.  The epilogue is wgrad_epi.h (slabs or atomics).
#include "conv_p.h"
#include "wgrad_epi.h"
#include <cstdlib>
#include <type_traits>

namespace evt_conv {
namespace {

__device__ __attribute__((aligned(256))) unsigned int g_zero_page_h[64];  // 256 zero bytes

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int G, int MAXA>
__device__ __forceinline__ void wait_groups(int ahead) {
  if constexpr (MAXA <= 0) { wait_vmcnt<0>(); }
  else {
    if (ahead >= MAXA) wait_vmcnt<(MAXA * G > 63 ? 63 : MAXA * G)>();
    else wait_groups<G, MAXA - 1>(ahead);
  }
}

// One transpose read, issued and NOT waited for (inline asm: through the builtin the compiler cannot tell the read from
// the LDS-DMA writes still in flight and drains vmcnt(0) before every stage, which serialises the ring).
template <int OFF>
__device__ __forceinline__ uint2 tr_rd(unsigned addr) {
  uint2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
// 8 consecutive positions of one channel = two reads (rows +0..3 and +4..7 of the lane group's 8-row band)
struct TrPair { uint2 lo, hi; };
__device__ __forceinline__ h16x8 tr_join(const TrPair& p) {
  union { uint4 u; h16x8 v; } r;
  r.u = make_uint4(p.lo.x, p.lo.y, p.hi.x, p.hi.y);
  return r.v;
}
// after the wait every consumer of the pair must come later: the empty volatile asm keeps its place behind the waitcnt asm
// and the MFMAs depend on its outputs
__device__ __forceinline__ void tr_tie(TrPair& p) { asm volatile("" : "+v"(p.lo), "+v"(p.hi)); }

// A-tile swizzle by row width: 256-byte rows (MA = 4), 128-byte rows (MA = 2) as in wgrad_ring; 64-byte rows (MA = 1: a
// 32-channel dy tile has the geometry of the window) as the window
template <int MA> __device__ __forceinline__ int a_swz(int row) {
  return MA == 4 ? 2 * ((row & 3) | (((row >> 3) & 1) << 2))
                 : (MA == 2 ? 2 * (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) : 2 * ((row >> 3) & 1));
}

constexpr int HPOS = 64;                       // positions per K stage
constexpr int HROWS = 128;                     // window rows staged per stage
constexpr int HWIN = HROWS * 64;               // 8 KiB

template <int MA, int NT, int NS>
__global__ __launch_bounds__(256) void wgrad_halo(WgP p, int spq, int stages_per_split) {
  constexpr int AROW = 64 * MA;                     // bytes per A-tile row (32*MA channels)
  constexpr int ABYTES = HPOS * AROW;
  constexpr int STAGE = ABYTES + HWIN;
  constexpr int G = MA + 2;                         // DMA instructions per wave per stage
  constexpr int RPI = 16 / MA;                      // A rows per DMA instruction
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j16 = lane & 15, g8 = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;

  const int ch = blockIdx.x % p.nchunk;
  const int a0 = (blockIdx.x / p.nchunk) * 32 * MA;

  const h16_t* Ag = reinterpret_cast<const h16_t*>(p.A);
  const h16_t* Bg = reinterpret_cast<const h16_t*>(p.B);
  const int nstages = p.nseq * spq;
  const int st_begin = blockIdx.y * stages_per_split;
  const int nst = min(nstages, st_begin + stages_per_split) - st_begin;
  if (nst <= 0) return;
  const int hrows = HPOS + (NT - 1) * p.dil;        // window rows in use

  const h16_t* zsrc = reinterpret_cast<const h16_t*>(g_zero_page_h);
  int arow[MA], acol[MA];
#pragma unroll
  for (int i = 0; i < MA; ++i) {
    arow[i] = wave * 16 + i * RPI + lane / (4 * MA);
    acol[i] = a0 + (((lane % (4 * MA)) ^ a_swz<MA>(arow[i])) * 8);
  }
  const int brow = wave * 16 + (lane >> 2);         // + 64 for the second pass: bit 3 of the row is the same
  const int bcol = ch * 32 + (((lane & 3) ^ (2 * ((brow >> 3) & 1))) * 8);

  auto issue = [&](int s) {
    unsigned char* base = smem + (s % NS) * STAGE;
    const int gs = st_begin + s;
    const int seq = gs / spq;
    const int q0 = (gs - seq * spq) * HPOS;
#pragma unroll
    for (int i = 0; i < MA; ++i) {
      const int q = q0 + arow[i];
      glds16(q < p.Q ? Ag + ((long)seq * p.LA + q) * p.CA + acol[i] : zsrc, base + wave * (16 * AROW) + i * 1024);
    }
    const h16_t* rsrc = Bg + ((long)seq * p.LB + q0 + p.off) * p.CB + bcol;
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int j = ps * 64 + brow;
      const int r = q0 + p.off + j;
      const bool ok = j < hrows && (unsigned)r < (unsigned)p.LB;
      glds16(ok ? rsrc + (long)j * p.CB : zsrc, base + ABYTES + ps * 4096 + wave * 1024);
    }
  };

  f32x4 acc[MA][NT];
#pragma unroll
  for (int i = 0; i < MA; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // dbias (column sums of the A operand = dy) by the chunk-0 blocks, from the staged tiles
  const bool do_bias = p.dbias != nullptr && ch == 0;                      // block-uniform
  f32x4 bacc[MA];
#pragma unroll
  for (int i = 0; i < MA; ++i) bacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const int frow = g8 * 8 + (j16 >> 2);
  const int fa = a_swz<MA>(frow);                   // unchanged by +4 and +32 rows
  const int half = (j16 & 1) * 8;
  const int cs = (j16 & 3) >> 1;
  int a_off[MA];
#pragma unroll
  for (int i = 0; i < MA; ++i) a_off[i] = frow * AROW + (((wr * 2 * MA + i * 2 + cs) ^ fa) * 16) + half;
  // window read offsets of every tap (ks = 0; ks = 1 adds 32 rows = 2048 bytes and leaves bit 3 of the row alone)
  int b_lo[NT], b_hi[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int r0 = frow + t * p.dil, r1 = r0 + 4;
    b_lo[t] = ABYTES + r0 * 64 + (((wc * 2 + cs) ^ (2 * ((r0 >> 3) & 1))) * 16) + half;
    b_hi[t] = ABYTES + r1 * 64 + (((wc * 2 + cs) ^ (2 * ((r1 >> 3) & 1))) * 16) + half;
  }

#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nst) issue(s);
  for (int s = 0; s < nst; ++s) {
    const int ahead = min(NS - 2, nst - 1 - s);
    wait_groups<G, NS - 2>(ahead);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (s + NS - 1 < nst) issue(s + NS - 1);
    const unsigned sbase = lds0 + (s % NS) * STAGE;
    TrPair fa[2][MA], fb[2][NT];                     // [1] unused without PIPE
    auto rd = [&](auto KS, TrPair (&a)[MA], TrPair (&b)[NT]) {
      constexpr int ks = decltype(KS)::value;
#pragma unroll
      for (int i = 0; i < MA; ++i) {
        a[i].lo = tr_rd<ks * 32 * AROW>(sbase + a_off[i]);
        a[i].hi = tr_rd<ks * 32 * AROW + 4 * AROW>(sbase + a_off[i]);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        b[t].lo = tr_rd<ks * 2048>(sbase + b_lo[t]);
        b[t].hi = tr_rd<ks * 2048>(sbase + b_hi[t]);
      }
    };
    auto landed = [&](TrPair (&a)[MA], TrPair (&b)[NT]) {
#pragma unroll
      for (int i = 0; i < MA; ++i) tr_tie(a[i]);
#pragma unroll
      for (int t = 0; t < NT; ++t) tr_tie(b[t]);
    };
    auto mma = [&](TrPair (&a)[MA], TrPair (&b)[NT]) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const h16x8 bv = tr_join(b[t]);
#pragma unroll
        for (int i = 0; i < MA; ++i) acc[i][t] = EVT_MFMA_16x16x32(tr_join(a[i]), bv, acc[i][t], 0, 0, 0);
      }
      if (do_bias) {
        h16x8 af[MA];
#pragma unroll
        for (int i = 0; i < MA; ++i) af[i] = tr_join(a[i]);
        wg_bias_mma<MA>(bacc, af, wc);
      }
    };
    // both halves' fragments live at once only where the register file allows two blocks per CU with them; no branch on
    // the sequence tail (its rows of A are zero): one straight-line body keeps the accumulators where they are
    constexpr bool PIPE = MA * NT * 4 + 2 * (MA + NT) * 4 <= 200;
    rd(std::integral_constant<int, 0>{}, fa[0], fb[0]);
    asm volatile("s_waitcnt lgkmcnt(0)");
    landed(fa[0], fb[0]);
    if constexpr (PIPE) {
      rd(std::integral_constant<int, 1>{}, fa[1], fb[1]);        // fly under the first half's MFMAs
      mma(fa[0], fb[0]);
      asm volatile("s_waitcnt lgkmcnt(0)");
      landed(fa[1], fb[1]);
      mma(fa[1], fb[1]);
    } else {
      mma(fa[0], fb[0]);
      rd(std::integral_constant<int, 1>{}, fa[0], fb[0]);
      asm volatile("s_waitcnt lgkmcnt(0)");
      landed(fa[0], fb[0]);
      mma(fa[0], fb[0]);
    }
    asm volatile("" ::: "memory");
  }
  if (do_bias) wg_finish_bias_mma<MA>(p, bacc, a0, wr, wc, g8, j16, blockIdx.y);
  wg_finish<MA, NT>(p, smem, acc, NT, a0, ch, 0, wr, wc, g8, j16, blockIdx.y);
}

template <int MA, int NT, int NS>
int launch_inst(const WgP& p, int spq, int per, hipStream_t st) {
  constexpr size_t lds = (size_t)NS * (HPOS * 64 * MA + HWIN);
  static_assert(lds >= 32 * (NT * 32 + 4) * 4, "epilogue scratch");
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_halo<MA, NT, NS>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return EVT_ELAUNCH;
    attr = true;
  }
  const long tiles = (long)(p.CA / (32 * MA)) * p.nchunk;
  evt_set_last_tag("wgrad_halo<bf16, %d, %dx32, 64, x%d>", 32 * MA, NT, NS);
  hipLaunchKernelGGL((wgrad_halo<MA, NT, NS>), dim3((unsigned)tiles, p.nsplit), dim3(256), lds, st, p, spq, per);
  return evt_check_launch();
}

}  // namespace

bool wgrad_halo_eligible(const WgP& p, int dtype) {
  static const bool off = getenv("EVT_NO_HALO") != nullptr;           // A/B switch for measurements
  if (off || dtype != EVT_DT_HALF) return false;
  if (p.s != 1 || p.KHp != p.KH) return false;
  if (p.KH != 3 && p.KH != 5 && p.KH != 7 && p.KH != 11) return false;
  if ((p.CA % 64 && p.CA != 32) || p.CB % 32) return false;
  if (p.Aact || p.Bact || p.a_slope != 1.f || p.b_slope != 1.f) return false;
  if (p.LA != p.Q) return false;
  if (HPOS + (p.KH - 1) * p.dil > HROWS) return false;
  // Measured at the B = 16 shapes (tools/bench_conv.py --wonly, us per launch: this kernel | wgrad_deep / wgrad_ring with
  // slabs | the same with atomics):  64 -> 64 k11 L5120: 29 | 46 | 44, k7: 23 | 34 | 32;  128 -> 128 k11 L2560: 46 | 40 | 65,
  // k7: 37 | 33 | 56;  256 -> 256 k11 L320: 30 | 27 | 35;  192 -> 384 k5 T200: 18 | 17 | 18;  1024 -> 1024 k5 H127: 238 | 125
  // | 139.  With 128 or more dy channels the 128 x 160 tile of wgrad_deep keeps two blocks per CU busy and wins; the
  // 64-channel tile here is bound by its LDS reads (52 transpose reads per 44 MFMAs).  So: 64 dy channels only
  // (EVT_HALO_ALL=1 lifts that for measurements).
  static const bool all = getenv("EVT_HALO_ALL") != nullptr;
  static const bool no32 = getenv("EVT_HALO_NO32") != nullptr;
  if (!all && p.CA != 64 && !(p.CA == 32 && !no32)) return false;
  if (p.CA % 128 == 0 && (long)(p.CA / 128) * (p.CB / 32) >= 128) return false;
  // sequence-local stages: the tail stage of a sequence is partly zero rows; DiscriminatorP's 23..127-long sequences
  // stay on the flat-position kernels
  const int spq = (p.Q + HPOS - 1) / HPOS;
  if ((long)spq * HPOS * 10 > (long)p.Q * 13) return false;            // <= 30 % zero rows in the sequence tails
  if ((long)p.nseq * spq < 8) return false;
  return true;
}

int launch_wgrad_halo(const WgP& p_in, hipStream_t st) {
  WgP p = p_in;
  if (!wgrad_halo_eligible(p, EVT_DT_HALF)) return EVT_ENOTSUP;
  p.nchunk = p.CB / 32;
  p.ntapgrp = 1;
  const int spq = (p.Q + HPOS - 1) / HPOS;
  const int nstages = p.nseq * spq;
  // 128-channel tiles when that still yields enough blocks, else 64
  const long tiles128 = p.CA % 128 == 0 ? (long)(p.CA / 128) * p.nchunk : 0;
  static const int force_ma = getenv("EVT_HALO_MA") ? atoi(getenv("EVT_HALO_MA")) : 0;
  int MA = tiles128 >= 64 ? 4 : 2;
  if (force_ma == 2 || (force_ma == 4 && tiles128 > 0)) MA = force_ma;
  if (p.CA == 32) MA = 1;                        // the 32-channel vocoder stage: one tile, all the parallelism from the split
  const long tiles = (long)(p.CA / (32 * MA)) * p.nchunk;
  static const long target = getenv("EVT_HALO_BLOCKS") ? atol(getenv("EVT_HALO_BLOCKS")) : 256;
  int nsplit, per;
  wgrad_pick_split(p, tiles, nstages, target, 3, &nsplit, &per);
  p.nsplit = nsplit;
  p.now_used = p.prev_used > nsplit ? p.prev_used : nsplit;
  if (p.parts > 0 && p.used_host) *p.used_host = p.now_used;
#define HALO(MA_, NS_)                                                                                   \
  (p.KH == 3 ? launch_inst<MA_, 3, NS_>(p, spq, per, st)                                                 \
             : p.KH == 5 ? launch_inst<MA_, 5, NS_>(p, spq, per, st)                                     \
                         : p.KH == 7 ? launch_inst<MA_, 7, NS_>(p, spq, per, st) : launch_inst<MA_, 11, NS_>(p, spq, per, st))
  return MA == 4 ? HALO(4, 3) : (MA == 2 ? HALO(2, 4) : HALO(1, 4));
#undef HALO
}

}  // namespace evt_conv
