// wn_layer: one layer of the WaveNet-style gated stack (WN, src/easevoice/module/modules.py:187-211 of the reference) --
//
//     x_in = in_layer(x)                      (k = 5, H -> 2H)          acts = tanh(x_in[:H] + g[:H]) * sigmoid(x_in[H:] + g[H:])
//     rs   = res_skip(acts)                   (1 x 1, H -> 2H; last layer H -> H)
//     x   <- (x + rs[:H]) * mask,  out <- out + rs[H:]                  (last layer: out <- (out + rs) * mask)
//
// -- forward, as ONE launch for the posterior encoder's / the flow's shape (H = 192, 16 x 200 = 3200 positions), gfx950.
// Unfused, a layer is four launches (conv_ring k = 5: 19 us, gate: 5 us, conv_ring 1 x 1: 14 us, residual: 5 us in the step)
// on a problem of 2.8 GFLOP and 6 MB: each of them is launch- and first-touch-bound, and the 32 layers of the two stacks are
// 1.4 ms of a 23 ms step.  Here the gate output never leaves the block before the 1 x 1 convolution reads it from LDS.
//
// The structure is resunit_wide's: one block = one tile of P = 16 NT positions of one sequence x ALL channels; four waves
// split the output channels; the activation rows sit in LDS (384-byte rows, 16-byte slots XOR-swizzled by the row so that the
// MFMA operand reads are conflict-free at every tap shift); every weight fragment is needed exactly once per block
// and goes from global memory (L2) straight into registers, R - 1 K steps ahead of its MFMAs.  The weights come in FRAGMENT
// ORDER (evt_frag_pack: [M-tile][K step][lane] x 16 bytes, a copy of the REG image made once per fold), so a wave's
// fragment load is 1 KiB contiguous: read from the REG image itself -- 16 rows x 64 bytes per instruction -- the same
// kernel ran at 12 B/clk per CU, 34-40 us per layer against 26 us for the four launches.  A wave owns the M-tiles
// {3 wm .. 3 wm + 2} of BOTH halves of a convolution's output, so a lane holds the tanh and the sigmoid argument (first
// convolution) or the residual and the skip term (second) of the same (channel, position): both epilogues are lane-local.
// With 4 waves x NT MFMAs per 1 KiB fragment the block is bound by its weight stream (884 KB per block and layer), which is
// why the ring is deeper than resunit_wide's and the grid (208 blocks of 16 positions) does not need to fill the chip.
// Measured (profiles/r05_wn_layer.txt): 18.0 us per layer against 25.5 us for the four launches, s2 step -0.34 ms; with the
// data half of the backward fused the same way (wn_layer_bwd below) 22.38-22.44 against 23.50-23.69 ms per step.
//
// Rounding points are those of the unfused launches (x_in, acts, rs are rounded to the 16-bit type where the separate
// kernels stored them), so the two paths differ only by the order of the fp32 sums inside a convolution.
// The tensors the backward needs (x_in, acts) are written as before; backward is unchanged (hip/wn.py).
#include "resunit_common.h"
#include "../../include/evt.h"
#include <cstdlib>
#include <type_traits>

namespace {

using namespace evt_ru;

struct WNP {
  const h16_t* x;        // [nseq][L][H] layer input
  const h16_t* w_in;     // in_layer in fragment order: [2H / 16 tiles][H / 32 chunks x K taps][64 lanes][8]
  const h16_t* w_rs;     // res_skip in fragment order: [2H / 16 (last: H / 16) tiles][H / 32][64][8]
  const float* b_in;     // [2H] or null
  const float* b_rs;     // [2H] (last: [H]) or null
  const h16_t* g;        // [nseq][2H] conditioning slice of this layer or null
  const h16_t* acc_in;   // [nseq][L][H] skip sum so far or null (first layer)
  const int* lens;       // [nseq] or null
  h16_t* x_in;           // out [nseq][L][2H]
  h16_t* acts;           // out [nseq][L][H]
  h16_t* x_out;          // out [nseq][L][H]; null when last
  h16_t* acc_out;        // out [nseq][L][H]
  int nseq, L, tps;
};

// the gate on the exp2 / rcp units: sigmoid(x) = 1 / (1 + 2^(-x log2 e)), tanh(x) = 2 sigmoid(2x) - 1 (absolute error
// < 3e-7: below half an ulp of the 16-bit result everywhere; saturates to 0 / 1 / +-1 through 2^(+-inf))
__device__ __forceinline__ float sigmoid_f(float x) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
__device__ __forceinline__ float tanh_f(float x) { return fmaf(2.f, sigmoid_f(2.f * x), -1.f); }

// 16-byte slot `slot` of row `row` inside a row of H * 2 bytes (H = 192: 24 slots, 384 bytes = 1.5 bank windows).  The MFMA B
// operand read is a ds_read_b128 of lane (n, g) at row r + n, slot 4 ch + g; the LDS serves it in four groups of 16 lanes
// that mix two values of g ({0-3, 12-15, 20-27}, ... : MI355X_MICROARCH.md, LDS), so "16 consecutive rows, one slot" is not
// the pattern to make conflict-free.  slot ^ (row & 7) is, for every row offset and tap shift (4 LDS cycles per read;
// tests/test_conv_index_math.py::test_wn_layer_lds_swizzles replays the bank arithmetic); the first version of this kernel
// used slot ^ ((row >> 1) & 7) and measured a third of its LDS cycles as conflicts (7 cycles per read).  The XOR stays inside
// an aligned group of eight slots (24 = 3 groups).
__device__ __forceinline__ int wslot(int row, int slot) { return slot ^ (row & 7); }

__device__ __forceinline__ void unpack4(const u32x2 v, float (&o)[4]) {
  o[0] = h2f_lo(v[0]); o[1] = h2f_hi(v[0]);
  o[2] = h2f_lo(v[1]); o[3] = h2f_hi(v[1]);
}

template <int H, int K, int NT, int R, bool LAST>
__global__ __launch_bounds__(256, 1) void wn_layer_fwd(WNP p) {
  constexpr int PITCH = H * 2;
  constexpr int SPR = H / 8;                 // 16-byte slots per row
  constexpr int NCH = H / 32;                // chunks of 32 input channels
  constexpr int MT = H / 16 / 4;             // M-tiles per wave and output half
  constexpr int P = 16 * NT;                 // positions per block
  constexpr int HALO = (K - 1) / 2;
  constexpr int XROWS = P + K - 1;
  constexpr int NKA = NCH * K, NKB = NCH;    // K steps of 32
  constexpr int MA = 2 * MT;                 // fragments per K step, first convolution
  constexpr int MB = LAST ? MT : 2 * MT;     // second
  static_assert(H % 64 == 0 && SPR % 8 == 0, "rows of whole 128-byte groups, M-tiles divisible by the four waves");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char* xs = smem;                                  // XROWS rows: x at positions q0 - HALO ..
  unsigned char* acl = smem + ((XROWS + 7) & ~7) * PITCH;     // P rows: acts at positions q0 ..
  const int tid = threadIdx.x, lane = tid & 63, wm = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int seq = blockIdx.x / p.tps;
  const int q0 = (blockIdx.x - seq * p.tps) * P;
  const long sbase = (long)seq * p.L;

  // ---- weight fragments: a ring of R sets per convolution, requested R - 1 K steps ahead ----
  u32x4 fa[R][MA];
  auto tileA = [&](int i) {         // M-tile of fragment i: tile (wm MT + i) of the first half, then of the second
    return i < MT ? wm * MT + i : H / 16 + wm * MT + i - MT;
  };
  auto issue_a = [&](const h16_t* w, const int nk, const int ks, const int s, auto cnt) {
    constexpr int CNT = decltype(cnt)::value;
#pragma unroll
    for (int i = 0; i < CNT; ++i)
      fa[s][i] = *reinterpret_cast<const u32x4*>(w + ((long)(tileA(i) * nk + ks) * 64 + lane) * 8);
  };
  using CA = std::integral_constant<int, MA>;
  using CB = std::integral_constant<int, MB>;

  // ---- every global read of the block that is not a weight is requested here, oldest first: the tile's rows (needed
  //      first: they go to LDS, zero outside the sequence), the first weight fragments, then what the two epilogues read
  //      (biases, conditioning, the skip sum so far) -- a load inside an epilogue is a serial round trip per tile ----
  constexpr int NP = XROWS * SPR;
  constexpr int PER = (NP + 255) / 256;
  uint4 xv[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int idx = u * 256 + tid;
    const int r = idx / SPR, sl = idx - r * SPR;
    const int pos = q0 - HALO + r;
    xv[u] = (idx < NP && pos >= 0 && pos < p.L) ? *reinterpret_cast<const uint4*>(p.x + (sbase + pos) * H + sl * 8)
                                                : make_uint4(0, 0, 0, 0);
  }
#pragma unroll
  for (int s = 0; s < R - 1 && s < NKA; ++s) issue_a(p.w_in, NKA, s, s, CA{});
  float bA0[MT][4], bA1[MT][4], bB0[MT][4], bB1[MT][4];
  u32x2 gA0[MT], gA1[MT], accp[MT][NT];
  const int len = p.lens ? p.lens[seq] : p.L;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int h = (wm * MT + i) * 16 + g * 4;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const f32x4 a0 = p.b_in ? *reinterpret_cast<const f32x4*>(p.b_in + h) : z;
    const f32x4 a1 = p.b_in ? *reinterpret_cast<const f32x4*>(p.b_in + H + h) : z;
    const f32x4 c0 = p.b_rs ? *reinterpret_cast<const f32x4*>(p.b_rs + h) : z;
    const f32x4 c1 = (!LAST && p.b_rs) ? *reinterpret_cast<const f32x4*>(p.b_rs + H + h) : z;
#pragma unroll
    for (int r = 0; r < 4; ++r) { bA0[i][r] = a0[r]; bA1[i][r] = a1[r]; bB0[i][r] = c0[r]; bB1[i][r] = c1[r]; }
    gA0[i] = gA1[i] = u32x2{0u, 0u};          // 16-bit zeros
    if (p.g) {
      gA0[i] = *reinterpret_cast<const u32x2*>(p.g + (long)seq * 2 * H + h);
      gA1[i] = *reinterpret_cast<const u32x2*>(p.g + (long)seq * 2 * H + H + h);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int pos = q0 + j * 16 + n;
      accp[i][j] = (p.acc_in && pos < p.L) ? *reinterpret_cast<const u32x2*>(p.acc_in + (sbase + pos) * H + h) : u32x2{0u, 0u};
    }
  }
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int idx = u * 256 + tid;
    if (idx >= NP) continue;
    const int r = idx / SPR, sl = idx - r * SPR;
    *reinterpret_cast<uint4*>(xs + r * PITCH + wslot(r, sl) * 16) = xv[u];
  }
  __syncthreads();

  // one convolution over NK K steps: acc[i][j] += W fragment i x rows of position tile j.  Fully unrolled, straight-line:
  // the compiler's wait counters then keep the younger weight requests in flight across a step.
  auto conv = [&](auto& acc, const h16_t* w, auto nk_c, auto taps_c, auto cnt, const unsigned char* rows) {
    constexpr int NK = decltype(nk_c)::value, TAPS = decltype(taps_c)::value, CNT = decltype(cnt)::value;
    u32x4 fb[NT];
    auto b_addr = [&](const int ks) {
      const int ch = ks / TAPS, tap = ks - ch * TAPS;
      const int row = n + tap;
      return rows + row * PITCH + wslot(row, ch * 4 + g) * 16;
    };
    {
      const unsigned char* b0 = b_addr(0);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        // rows 16 j + n + tap: 16 j rows further is 16 j * PITCH bytes further only if the swizzle agrees -- row & 7 repeats
        // every 8 rows, so it does
        fb[j] = *reinterpret_cast<const u32x4*>(b0 + j * 16 * PITCH);
      }
    }
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
      const int s = ks % R;
      if (ks + R - 1 < NK) issue_a(w, NK, ks + R - 1, (ks + R - 1) % R, cnt);
#pragma unroll
      for (int i = 0; i < CNT; ++i) tie(fa[s][i]);
      const unsigned char* nb = b_addr(ks + 1 < NK ? ks + 1 : ks);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        tie(fb[j]);
#pragma unroll
        for (int i = 0; i < CNT; ++i)
          acc[i][j] = EVT_MFMA_16x16x32(as_h8(fa[s][i]), as_h8(fb[j]), acc[i][j], 0, 0, 0);
        fb[j] = *reinterpret_cast<const u32x4*>(nb + j * 16 * PITCH);
      }
    }
  };

  // ---- first convolution (k = K) + gate ----
  {
    f32x4 acc[MA][NT];
#pragma unroll
    for (int i = 0; i < MA; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    conv(acc, p.w_in, std::integral_constant<int, NKA>{}, std::integral_constant<int, K>{}, CA{}, xs);
    // the second convolution's first weights travel under this epilogue
#pragma unroll
    for (int s = 0; s < R - 1 && s < NKB; ++s) issue_a(p.w_rs, NKB, s, s, CB{});
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int h = (wm * MT + i) * 16 + g * 4;
      float ga[4], gb[4];
      unpack4(gA0[i], ga);
      unpack4(gA1[i], gb);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int m = j * 16 + n;
        const int pos = q0 + m;
        h16_t xa[4], xb[4], o4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          xa[r] = f2h(acc[i][j][r] + bA0[i][r]);
          xb[r] = f2h(acc[MT + i][j][r] + bA1[i][r]);
          const float a = h2f(xa[r]) + ga[r], b = h2f(xb[r]) + gb[r];
          o4[r] = f2h(tanh_f(a) * sigmoid_f(b));
        }
        *reinterpret_cast<uint2*>(acl + m * PITCH + wslot(m, h >> 3) * 16 + (h & 7) * 2) = *reinterpret_cast<uint2*>(o4);
        if (pos < p.L) {
          h16_t* xi = p.x_in + (sbase + pos) * 2 * H + h;
          *reinterpret_cast<uint2*>(xi) = *reinterpret_cast<uint2*>(xa);
          *reinterpret_cast<uint2*>(xi + H) = *reinterpret_cast<uint2*>(xb);
          *reinterpret_cast<uint2*>(p.acts + (sbase + pos) * H + h) = *reinterpret_cast<uint2*>(o4);
        }
      }
    }
  }
  __syncthreads();

  // ---- second convolution (1 x 1) + residual / skip ----
  {
    f32x4 acc[MB][NT];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    conv(acc, p.w_rs, std::integral_constant<int, NKB>{}, std::integral_constant<int, 1>{}, CB{}, acl);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int h = (wm * MT + i) * 16 + g * 4;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int m = j * 16 + n;
        const int pos = q0 + m;
        if (pos >= p.L) continue;
        const bool live = pos < len;
        float ap[4];
        unpack4(accp[i][j], ap);
        h16_t o1[4], o2[4];
        if constexpr (LAST) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float rs = h2f(f2h(acc[i][j][r] + bB0[i][r]));
            o2[r] = f2h(live ? ap[r] + rs : 0.f);
          }
        } else {
          float xo[4];
          const int xr = HALO + m;
          unpack4(*reinterpret_cast<const u32x2*>(xs + xr * PITCH + wslot(xr, h >> 3) * 16 + (h & 7) * 2), xo);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float r0 = h2f(f2h(acc[i][j][r] + bB0[i][r]));
            const float r1 = h2f(f2h(acc[MT + i][j][r] + bB1[i][r]));
            o1[r] = f2h(live ? xo[r] + r0 : 0.f);
            o2[r] = f2h(ap[r] + r1);
          }
          *reinterpret_cast<uint2*>(p.x_out + (sbase + pos) * H + h) = *reinterpret_cast<uint2*>(o1);
        }
        *reinterpret_cast<uint2*>(p.acc_out + (sbase + pos) * H + h) = *reinterpret_cast<uint2*>(o2);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The DATA half of the layer's backward in one launch (instead of evt_wn_residual_bwd, the 1 x 1 backward-data launch,
// evt_gated_act_bwd and the k = 5 backward-data launch with its add epilogue):
//
//     drs    = [dx_next * mask | dacc]                     (last layer: drs = dacc * mask, H channels)
//     dacts  = res_skip^T drs                              (1 x 1: H rows x 2H, ALT image)
//     dx_in  = gate'(x_in + g) dacts                       (tanh / sigmoid derivatives; dg[seq] += sum over positions)
//     dx     = in_layer^T dx_in + dx_next * mask           (k = 5, ALT image = flipped taps: a plain correlation)
//
// Same structure as the forward: "first convolution on tile + halo into LDS, second convolution on the tile"; the rows are
// 2H wide here (768 bytes: wslot2).  A wave owns M-tiles {3 wm ..
// 3 wm + 2} of the H output rows of both convolutions: the lane that holds dacts of (channel h, position) writes both gate
// gradients, channels h and H + h of dx_in.  drs and dx_in are also written to global memory: the two weight-gradient
// launches (side stream) read them, as before.
struct WNB {
  const h16_t* dx_next;  // [nseq][L][H] gradient of the layer's x output; null: none (last layer)
  const h16_t* dacc;     // [nseq][L][H] gradient of the skip sum
  const h16_t* x_in;     // [nseq][L][2H] saved by the forward
  const h16_t* g;        // [nseq][2H] or null
  const h16_t* w_rs;     // ALT image of res_skip, fragment order: H / 16 tiles x (2H or H) / 32 K steps
  const h16_t* w_in;     // ALT image of in_layer, fragment order: H / 16 tiles x (2H / 32 chunks x K taps)
  const int* lens;
  h16_t* drs;            // out [nseq][L][2H] (last: [nseq][L][H])
  h16_t* dx_in;          // out [nseq][L][2H]
  h16_t* dx;             // out [nseq][L][H]
  float* dg;             // [nseq][2H] fp32, += (atomics), or null
  int nseq, L, tps;
};

// 768-byte rows (3 bank windows): slot ^ ((row & 7) << 1) is conflict-free for the same read (see wslot); the XOR stays inside an
// aligned group of 16 slots (48 = 3 groups)
__device__ __forceinline__ int wslot2(int row, int slot) { return slot ^ ((row & 7) << 1); }

// sum over the 16 lanes that share g (the positions n of one tile): xor-butterfly inside a row of 16 lanes
__device__ __forceinline__ float sum16(float v) {
  v += __shfl_xor(v, 1, 16);
  v += __shfl_xor(v, 2, 16);
  v += __shfl_xor(v, 4, 16);
  v += __shfl_xor(v, 8, 16);
  return v;
}

template <int H, int K, int NT, int R, bool LAST>
__global__ __launch_bounds__(256, 1) void wn_layer_bwd(WNB p) {
  constexpr int PB = H * 4;                  // row pitch: 2H channels
  constexpr int SPRB = H / 4;                // 16-byte slots per row
  constexpr int MT = H / 16 / 4;             // M-tiles per wave
  constexpr int P = 16 * NT;                 // own positions per block
  constexpr int HALO = (K - 1) / 2;
  constexpr int NT1 = NT + 1;                // position tiles of the first convolution: P + 2 HALO <= 16 NT1 rows
  constexpr int ROWS = P + 2 * HALO;         // rows that matter in both LDS regions
  constexpr int CIN1 = LAST ? H : 2 * H;     // channels of drs
  constexpr int NK1 = CIN1 / 32, NK2 = (2 * H / 32) * K;
  static_assert(2 * HALO <= 16, "one extra position tile covers the halo");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char* ds = smem;                          // 16 NT1 rows: drs at positions q0 - HALO ..
  unsigned char* es = smem + 16 * NT1 * PB;          // 16 NT1 rows: dx_in at positions q0 - HALO ..
  const int tid = threadIdx.x, lane = tid & 63, wm = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int seq = blockIdx.x / p.tps;
  const int q0 = (blockIdx.x - seq * p.tps) * P;
  const long sbase = (long)seq * p.L;
  const int len = p.lens ? p.lens[seq] : p.L;

  u32x4 fa[R][MT];
  auto issue_a = [&](const h16_t* w, const int nk, const int ks, const int s) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
      fa[s][i] = *reinterpret_cast<const u32x4*>(w + ((long)((wm * MT + i) * nk + ks) * 64 + lane) * 8);
  };

  // ---- every non-weight global read up front (oldest first): the drs rows, then what the gate epilogue reads ----
  constexpr int SPR1 = CIN1 / 8;             // 16-byte pieces per staged row
  constexpr int NP = ROWS * SPR1;
  constexpr int PER = (NP + 255) / 256;
  uint4 dv[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int idx = u * 256 + tid;
    const int r = idx / SPR1, sl = idx - r * SPR1;
    const int pos = q0 - HALO + r;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (idx < NP && pos >= 0 && pos < p.L) {
      if (LAST) {
        if (pos < len) v = *reinterpret_cast<const uint4*>(p.dacc + (sbase + pos) * H + sl * 8);
      } else if (sl < H / 8) {
        if (p.dx_next && pos < len) v = *reinterpret_cast<const uint4*>(p.dx_next + (sbase + pos) * H + sl * 8);
      } else {
        v = *reinterpret_cast<const uint4*>(p.dacc + (sbase + pos) * H + (sl - H / 8) * 8);
      }
    }
    dv[u] = v;
  }
#pragma unroll
  for (int s = 0; s < R - 1 && s < NK1; ++s) issue_a(p.w_rs, NK1, s, s);
  u32x2 xa[MT][NT1], xb[MT][NT1], ga[MT], gb[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int h = (wm * MT + i) * 16 + g * 4;
    ga[i] = gb[i] = u32x2{0u, 0u};
    if (p.g) {
      ga[i] = *reinterpret_cast<const u32x2*>(p.g + (long)seq * 2 * H + h);
      gb[i] = *reinterpret_cast<const u32x2*>(p.g + (long)seq * 2 * H + H + h);
    }
#pragma unroll
    for (int j = 0; j < NT1; ++j) {
      const int m = j * 16 + n;
      const int pos = q0 - HALO + m;
      const bool in = m < ROWS && pos >= 0 && pos < p.L;
      xa[i][j] = in ? *reinterpret_cast<const u32x2*>(p.x_in + (sbase + pos) * 2 * H + h) : u32x2{0u, 0u};
      xb[i][j] = in ? *reinterpret_cast<const u32x2*>(p.x_in + (sbase + pos) * 2 * H + H + h) : u32x2{0u, 0u};
    }
  }
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int idx = u * 256 + tid;
    if (idx >= NP) continue;
    const int r = idx / SPR1, sl = idx - r * SPR1;
    *reinterpret_cast<uint4*>(ds + r * PB + wslot2(r, sl) * 16) = dv[u];
    const int pos = q0 - HALO + r;
    if (r >= HALO && r < HALO + P && pos < p.L)            // own rows: drs for the res_skip weight gradient
      *reinterpret_cast<uint4*>(p.drs + (sbase + pos) * CIN1 + sl * 8) = dv[u];
  }
  __syncthreads();

  auto conv = [&](auto& acc, const h16_t* w, auto nk_c, auto taps_c, auto nt_c, const unsigned char* rows) {
    constexpr int NK = decltype(nk_c)::value, TAPS = decltype(taps_c)::value, NTC = decltype(nt_c)::value;
    u32x4 fb[NTC];
    auto b_addr = [&](const int ks) {
      const int ch = ks / TAPS, tap = ks - ch * TAPS;
      const int row = n + tap;
      return rows + row * PB + wslot2(row, ch * 4 + g) * 16;
    };
    {
      const unsigned char* b0 = b_addr(0);
#pragma unroll
      for (int j = 0; j < NTC; ++j) fb[j] = *reinterpret_cast<const u32x4*>(b0 + j * 16 * PB);   // (row + 16) & 7 == row & 7
    }
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
      const int s = ks % R;
      if (ks + R - 1 < NK) issue_a(w, NK, ks + R - 1, (ks + R - 1) % R);
#pragma unroll
      for (int i = 0; i < MT; ++i) tie(fa[s][i]);
      const unsigned char* nb = b_addr(ks + 1 < NK ? ks + 1 : ks);
#pragma unroll
      for (int j = 0; j < NTC; ++j) {
        tie(fb[j]);
#pragma unroll
        for (int i = 0; i < MT; ++i)
          acc[i][j] = EVT_MFMA_16x16x32(as_h8(fa[s][i]), as_h8(fb[j]), acc[i][j], 0, 0, 0);
        fb[j] = *reinterpret_cast<const u32x4*>(nb + j * 16 * PB);
      }
    }
  };

  // ---- first convolution (1 x 1 transposed) on tile + halo, gate derivative ----
  {
    f32x4 acc[MT][NT1];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT1; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    conv(acc, p.w_rs, std::integral_constant<int, NK1>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, NT1>{}, ds);
#pragma unroll
    for (int s = 0; s < R - 1 && s < NK2; ++s) issue_a(p.w_in, NK2, s, s);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int h = (wm * MT + i) * 16 + g * 4;
      float gaf[4], gbf[4], sa[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
      unpack4(ga[i], gaf);
      unpack4(gb[i], gbf);
#pragma unroll
      for (int j = 0; j < NT1; ++j) {
        const int m = j * 16 + n;
        const int pos = q0 - HALO + m;
        const bool in = m < ROWS && pos >= 0 && pos < p.L;
        const bool own = in && m >= HALO && m < HALO + P;
        float xaf[4], xbf[4];
        unpack4(xa[i][j], xaf);
        unpack4(xb[i][j], xbf);
        h16_t oa[4], ob[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = h2f(f2h(acc[i][j][r]));            // dacts as the 1 x 1 launch stored it
          const float t = tanh_f(xaf[r] + gaf[r]), sg = sigmoid_f(xbf[r] + gbf[r]);
          const float da = in ? d * sg * (1.f - t * t) : 0.f, db = in ? d * t * sg * (1.f - sg) : 0.f;
          oa[r] = f2h(da);
          ob[r] = f2h(db);
          if (own) { sa[r] += da; sb[r] += db; }
        }
        unsigned char* ep = es + m * PB;
        *reinterpret_cast<uint2*>(ep + wslot2(m, h >> 3) * 16 + (h & 7) * 2) = *reinterpret_cast<uint2*>(oa);
        *reinterpret_cast<uint2*>(ep + wslot2(m, (H + h) >> 3) * 16 + (h & 7) * 2) = *reinterpret_cast<uint2*>(ob);
        if (own) {
          h16_t* o = p.dx_in + (sbase + pos) * 2 * H + h;
          *reinterpret_cast<uint2*>(o) = *reinterpret_cast<uint2*>(oa);
          *reinterpret_cast<uint2*>(o + H) = *reinterpret_cast<uint2*>(ob);
        }
      }
      if (p.dg) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float ta = sum16(sa[r]), tb = sum16(sb[r]);
          if (n == 0) {
            atomicAdd(p.dg + (long)seq * 2 * H + h + r, ta);
            atomicAdd(p.dg + (long)seq * 2 * H + H + h + r, tb);
          }
        }
      }
    }
  }
  __syncthreads();

  // ---- second convolution (k = K transposed: the ALT image holds the flipped taps) + the residual branch's gradient ----
  {
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    conv(acc, p.w_in, std::integral_constant<int, NK2>{}, std::integral_constant<int, K>{}, std::integral_constant<int, NT>{}, es);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int h = (wm * MT + i) * 16 + g * 4;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int m = j * 16 + n;
        const int pos = q0 + m;
        if (pos >= p.L) continue;
        float dr[4] = {0.f, 0.f, 0.f, 0.f};
        if (!LAST) {
          const int r0 = HALO + m;                             // drs[:H] of this position = dx_next * mask
          unpack4(*reinterpret_cast<const u32x2*>(ds + r0 * PB + wslot2(r0, h >> 3) * 16 + (h & 7) * 2), dr);
        }
        h16_t o4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o4[r] = f2h(acc[i][j][r] + dr[r]);
        *reinterpret_cast<uint2*>(p.dx + (sbase + pos) * H + h) = *reinterpret_cast<uint2*>(o4);
      }
    }
  }
}

// REG image [rows][ktot] -> fragment order [rows / 16][ktot / 32][64 lanes][8]: lane (n = lane & 15, g = lane >> 4) of
// (tile, K step) holds row 16 tile + n, K elements 32 ks + 8 g .. + 7 -- the MFMA A operand of that step as one 16-byte load
__global__ __launch_bounds__(256) void frag_pack_kernel(const evt_frag_item* items) {
  const evt_frag_item it = items[blockIdx.y];
  const int nk = it.ktot / 32;
  const long pieces = (long)it.rows * it.ktot / 8;
  const uint4* src = reinterpret_cast<const uint4*>(it.src);
  uint4* dst = reinterpret_cast<uint4*>(it.dst);
  for (long pc = blockIdx.x * 256L + threadIdx.x; pc < pieces; pc += gridDim.x * 256L) {
    const int lane = (int)(pc & 63);
    const long ts = pc >> 6;
    const int tile = (int)(ts / nk), ks = (int)(ts - (long)tile * nk);
    dst[pc] = src[((long)(tile * 16 + (lane & 15)) * it.ktot + ks * 32 + (lane >> 4) * 8) / 8];
  }
}

template <int NT, int R, bool LAST>
int launch(WNP p, hipStream_t st) {
  constexpr int H = 192, K = 5, P = 16 * NT;
  p.tps = (p.L + P - 1) / P;
  const size_t lds = (size_t)(((P + K - 1 + 7) & ~7) + P) * H * 2;
  evt_set_last_tag("wn_layer_fwd<%s, %d, k%d, nt %d, ring %d%s>", EVT_HALF_NAME, H, K, NT, R, LAST ? ", last" : "");
  hipLaunchKernelGGL((wn_layer_fwd<H, K, NT, R, LAST>), dim3(p.nseq * p.tps), dim3(256), lds, st, p);
  return evt_check_launch();
}

template <bool LAST>
int launch_nt(WNP p, hipStream_t st) {
  // positions per block (16 NT) and ring depth, measured on the B = 16 x 200 shape as a replayed graph of 16 layers with
  // every layer's weights a first touch (tools/bench_wn.py --flush; profiles/r05_wn_layer.txt), us per layer:
  //   NT = 1 (208 blocks): 18.0   NT = 2 (112 blocks): 20.0   NT = 3 (80 blocks): 35 (before the epilogue loads were hoisted)
  //   ring 4 and 8 alike; the four launches this replaces: 25.5.  In the s2 step: 22.96-22.98 ms (NT = 1), 23.10-23.18
  //   (NT = 2), 23.29-23.33 with the four launches.  More positions per block do not pay: a block is bound by its own
  //   serial chain (stage rows, 884 KB of weights at the CU's 64 B/clk, two epilogues), and the chip has CUs to spare.
  static const int nt = getenv("EVT_WN_NT") ? atoi(getenv("EVT_WN_NT")) : 1;
  static const int ring = getenv("EVT_WN_RING") ? atoi(getenv("EVT_WN_RING")) : 8;
  if (ring <= 4) {
    if (nt == 1) return launch<1, 4, LAST>(p, st);
    if (nt == 3) return launch<3, 4, LAST>(p, st);
    return launch<2, 4, LAST>(p, st);
  }
  if (nt == 1) return launch<1, 8, LAST>(p, st);
  if (nt == 3) return launch<3, 8, LAST>(p, st);
  return launch<2, 8, LAST>(p, st);
}

template <int NT, int R, bool LAST>
int launch_b(WNB p, hipStream_t st) {
  constexpr int H = 192, K = 5, P = 16 * NT;
  p.tps = (p.L + P - 1) / P;
  const size_t lds = (size_t)2 * 16 * (NT + 1) * H * 4;
  static bool attr = false;
  if (!attr && lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wn_layer_bwd<H, K, NT, R, LAST>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return EVT_ELAUNCH;
    attr = true;
  }
  evt_set_last_tag("wn_layer_bwd<%s, %d, k%d, nt %d, ring %d%s>", EVT_HALF_NAME, H, K, NT, R, LAST ? ", last" : "");
  hipLaunchKernelGGL((wn_layer_bwd<H, K, NT, R, LAST>), dim3(p.nseq * p.tps), dim3(256), lds, st, p);
  return evt_check_launch();
}

template <bool LAST>
int launch_b_nt(WNB p, hipStream_t st) {
  // 16 or 32 own positions per block: 22.38 / 22.44 vs 22.44 / 22.37 ms per s2 step -- alike; 16 keeps the LDS under 64 KB
  static const int nt = getenv("EVT_WN_BWD_NT") ? atoi(getenv("EVT_WN_BWD_NT")) : 1;
  if (nt == 2) return launch_b<2, 8, LAST>(p, st);
  return launch_b<1, 8, LAST>(p, st);
}

}  // namespace

extern "C" {

int32_t evt_wn_layer_supported(int32_t dtype, int32_t H, int32_t k, int32_t dil) {
  static const bool off = getenv("EVT_NO_WN_LAYER") != nullptr;      // A/B switch for measurements
  return (!off && dtype == EVT_DT_HALF && H == 192 && k == 5 && dil == 1) ? 1 : 0;
}

int evt_frag_pack(const evt_frag_item* items, int32_t nitems, void* stream) {
  if (!items || nitems <= 0) return EVT_EINVAL;
  evt_set_last_tag("frag_pack x%d", nitems);
  hipLaunchKernelGGL(frag_pack_kernel, dim3(32, nitems), dim3(256), 0, (hipStream_t)stream, items);
  return evt_check_launch();
}

int evt_wn_layer_fwd(int32_t dtype, const void* x, const void* w_in_frag, const float* b_in, const void* w_rs_frag,
                     const float* b_rs, const void* g, const void* acc_in, const int32_t* lens, void* x_in, void* acts,
                     void* x_out, void* acc_out, int32_t nseq, int32_t L, int32_t H, int32_t k, int32_t last, void* stream) {
  if (!evt_wn_layer_supported(dtype, H, k, 1)) return EVT_ENOTSUP;
  if (!x || !w_in_frag || !w_rs_frag || !x_in || !acts || !acc_out || (!last && !x_out) || nseq <= 0 || L <= 0) return EVT_EINVAL;
  WNP p{};
  p.x = (const h16_t*)x; p.w_in = (const h16_t*)w_in_frag; p.w_rs = (const h16_t*)w_rs_frag; p.b_in = b_in; p.b_rs = b_rs;
  p.g = (const h16_t*)g; p.acc_in = (const h16_t*)acc_in; p.lens = lens;
  p.x_in = (h16_t*)x_in; p.acts = (h16_t*)acts; p.x_out = (h16_t*)x_out; p.acc_out = (h16_t*)acc_out;
  p.nseq = nseq; p.L = L;
  hipStream_t st = (hipStream_t)stream;
  return last ? launch_nt<true>(p, st) : launch_nt<false>(p, st);
}

int evt_wn_layer_bwd_data(int32_t dtype, const void* dx_next, const void* dacc, const void* x_in, const void* g,
                          const void* w_rs_alt_frag, const void* w_in_alt_frag, const int32_t* lens, void* drs, void* dx_in,
                          void* dx, float* dg, int32_t nseq, int32_t L, int32_t H, int32_t k, int32_t last, void* stream) {
  if (!evt_wn_layer_supported(dtype, H, k, 1)) return EVT_ENOTSUP;
  if (!dacc || !x_in || !w_rs_alt_frag || !w_in_alt_frag || !drs || !dx_in || !dx || nseq <= 0 || L <= 0) return EVT_EINVAL;
  if (last && dx_next) return EVT_EINVAL;          // the last layer has no x output
  WNB p{};
  p.dx_next = (const h16_t*)dx_next; p.dacc = (const h16_t*)dacc; p.x_in = (const h16_t*)x_in; p.g = (const h16_t*)g;
  p.w_rs = (const h16_t*)w_rs_alt_frag; p.w_in = (const h16_t*)w_in_alt_frag; p.lens = lens;
  p.drs = (h16_t*)drs; p.dx_in = (h16_t*)dx_in; p.dx = (h16_t*)dx; p.dg = dg;
  p.nseq = nseq; p.L = L;
  hipStream_t st = (hipStream_t)stream;
  return last ? launch_b_nt<true>(p, st) : launch_b_nt<false>(p, st);
}

}  // extern "C"
