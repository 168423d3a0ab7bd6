// gemm256: 256 x 256 x 64 bf16 GEMM for the dense layers of the s1 transformer (gfx950), forward and backward-data.
//
// Reference call sites: F.linear of the packed in-projection / out-projection (patched_mha_with_cache.py:242,460) and
// linear1 / linear2 (transformer.py:207-224,330-334) at M = B x L = 32768 rows, K, N in {512, 1536, 2048}: 16 of the 26 ms
// of GEMM time per micro-step.  The 128 x 128 kernels of conv_deep.hip move (128 + 128) x K operand bytes through LDS-DMA
// per 128 x 128 outputs; at K = 512 a block lives for 8-16 stages, so fill / drain and operand traffic, not MFMA rate,
// set their 560-600 TFLOP/s.  This kernel halves the traffic per flop and keeps three operand pieces in flight:
//
//   out[M][NO] = epilogue( B[M][K] . A[NO][K]^T )      A = weight rows (REG image: forward; ALT image = W^T: backward-data)
//
// Block = 256 (NO) x 256 (M) outputs, 8 waves as 2 x 4, a wave owns 128 x 64 = 8 x 4 MFMA tiles (mfma_f32_16x16x32_bf16,
// A operand = weight rows, B operand = token rows: a lane ends up with 4 consecutive output channels of one token).
// One K tile (64 wide) is FOUR 16 KiB pieces -- a0 / a1: the weight rows the first / second half of every wave's M-tiles
// read, b0 / b1 likewise for the token rows -- so that each of the four phases of a K tile (one 64 x 32 quadrant of the
// wave's accumulators = 16 MFMAs) reads ONE new piece (phase 0: a0 + b0, 1: b1, 2: a1, 3: b0 again) and every LDS region
// is dead a known number of phases after it was filled.  Pieces are written by LDS-DMA (global_load_lds_dwordx4, the
// bank swizzle on the per-lane source address, undone on the ds_read_b128), one piece per phase, FIVE pieces ahead of
// the phase that issues it: the piece overwritten was last read two phases ago, the piece needed next was issued four
// phases ago.  Waits are counted (s_waitcnt vmcnt(6): three pieces stay in flight), one raw s_barrier per phase.
// Two K-tile buffers = 128 KiB LDS, one block per CU.
//
// Epilogue (all optional): + bias, relu, dropout (the counter hash of enc_ops.hip on the flat output index), gate
// (x gate_pos where gate > 0, else 0: the relu + dropout derivative read off the saved activation), + add (the residual
// branch's gradient) -- what the s1 block otherwise runs as separate element-wise launches.
#include "evt_common.h"
#include "../../include/evt.h"
#include <type_traits>
#include <cstdlib>

namespace {

__device__ __attribute__((aligned(256))) unsigned int g256_zero_page[64];  // 256 zero bytes

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct G256 {
  const h16_t* a;      // [NO][K] weight-side rows
  const h16_t* b;      // [M][K] token-side rows
  h16_t* out;          // [M][NO]
  const float* bias;    // [NO] or null
  const h16_t* gate;   // [M][NO] or null
  const h16_t* add;    // [M][NO] or null
  const unsigned* seed_dev;
  int M, NO, K;
  int relu;
  unsigned thr, site;   // dropout: keep iff hash >= thr (0 = off)
  float keep, gate_pos;
  int P, Y;             // token tiles, channel tiles
};

constexpr int PIECE = 128 * 128;       // bytes: 128 rows of 64 bf16
constexpr int KTB = 4 * PIECE;         // one K tile: regions a0 | b0 | b1 | a1
constexpr int R_A0 = 0, R_B0 = 1, R_B1 = 2, R_A1 = 3;
constexpr int EP_PITCH = 272;            // epilogue staging: a wave's 64 token rows x 128 channels (256 B) + 16 B pad
constexpr int EP_WAVE = 64 * EP_PITCH;   // 17 KiB per wave, 136 KiB per block
constexpr int LDS_MAIN = 8 * EP_WAVE > 2 * KTB ? 8 * EP_WAVE : 2 * KTB;
constexpr int LDS_TOTAL = LDS_MAIN + 256 * 4;     // + the tile's bias values

__device__ __forceinline__ unsigned mix32(unsigned x) {   // lowbias32 finaliser (enc_ops.hip)
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// one quadrant of the wave's accumulators: M-tiles 4*QA .. 4*QA+3, N-tiles 2*QB, 2*QB+1, both k sub-steps
template <int QA, int QB>
__device__ __forceinline__ void mma_quadrant(f32x4 (&acc)[8][4], const h16x8 (&af)[4][2], const h16x8 (&bfr)[2][2]) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[QA * 4 + i][QB * 2 + j] =
            EVT_MFMA_16x16x32(af[i][ks], bfr[j][ks], acc[QA * 4 + i][QB * 2 + j], 0, 0, 0);
}

// ablation variants: the fragment registers stay live (and their ds_reads issued) without the matrix pipe
__device__ __forceinline__ void keep_frags(const h16x8 (&af)[4][2], const h16x8 (&bfr)[2][2]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) { asm volatile("" ::"v"(af[i][0])); asm volatile("" ::"v"(af[i][1])); }
#pragma unroll
  for (int j = 0; j < 2; ++j) { asm volatile("" ::"v"(bfr[j][0])); asm volatile("" ::"v"(bfr[j][1])); }
}

// VAR: measurement variants (tools/bench_gemm256.py; the product launches VAR = 0): bit 0 no MFMAs, bit 1 no DMA after the
// prologue, bit 2 no fragment reads, bit 3 no epilogue stores
template <int VAR>
__global__ __launch_bounds__(512, 2) void gemm256_nt(G256 p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int wr = wave >> 2, wc = wave & 3;

  // Persistent blocks: one per CU, walking the tile list with stride gridDim.x (a multiple of 8: a block stays on its XCD).
  // XCD-aware decode (as conv_deep): all channel tiles of a token tile on one XCD, so the token rows are fetched once per
  // L2.  Being persistent is what hides the output: the stores of tile t drain under the DMA and MFMAs of tile t + 1 (one
  // block per CU could not overlap them with anything: 91 us instead of 48 us without stores at [32768, 1536, 512]).
  const int nvirt = 8 * ((p.P + 7) / 8) * p.Y;
  for (int lin = blockIdx.x; lin < nvirt; lin += gridDim.x) {
  const int xcd = lin & 7, slot = lin >> 3;
  const int yi = slot % p.Y;
  const int pb = xcd + 8 * (slot / p.Y);
  if (pb >= p.P) continue;

  // ---- per-lane DMA sources.  A wave fills local rows [16 w, 16 w + 16) of every piece, 8 rows per instruction.
  //      piece-local row lr -> tile row:  a-piece h: (lr >> 6) * 128 + h * 64 + (lr & 63)   (wave row-half, M-tile half)
  //                                       b-piece h: (lr >> 5) * 64 + h * 32 + (lr & 31)    (wave column, N-tile half)
  const int rsub = lane >> 3, pslot = lane & 7;
  unsigned offs[4][2];          // element offsets of this lane's 16 bytes at K tile 0, per region and instruction
  bool bok[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int lr = wave * 16 + i * 8 + rsub;
    const int c = pslot ^ (lr & 7);                       // logical 16-byte k-slot this lane fetches
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int ar = yi * 256 + (lr >> 6) * 128 + h * 64 + (lr & 63);
      offs[h ? R_A1 : R_A0][i] = (unsigned)ar * (unsigned)p.K + c * 8;
      const int br = pb * 256 + (lr >> 5) * 64 + h * 32 + (lr & 31);
      bok[h][i] = br < p.M;
      offs[h ? R_B1 : R_B0][i] = (unsigned)(bok[h][i] ? br : 0) * (unsigned)p.K + c * 8;
    }
  }
  // the tile's 256 bias values wait in LDS behind the staging area (read in the epilogue: per-lane global loads of them
  // there were serialised round trips, ~40 % of the kernel's time)
  float* bias_l = reinterpret_cast<float*>(smem + LDS_MAIN);
  if (p.bias && wave < 4)      // LDS-DMA like the operands (a register load + ds_write would stall the tile's start)
    __builtin_amdgcn_global_load_lds((gptr_t)(p.bias + yi * 256 + wave * 64 + lane), (lptr_t)(bias_l + wave * 64), 4, 0, 0);
  const h16_t* zsrc = reinterpret_cast<const h16_t*>(g256_zero_page) + pslot * 8;
  unsigned char* my = smem + wave * 2048;                  // + buf * KTB + region * PIECE + i * 1024
  const int nt = p.K >> 6;

  // piece (kt, region): two DMA instructions per wave
  auto issue_a = [&](int kt, int region) {
    unsigned char* dst = my + (kt & 1) * KTB + region * PIECE;
    if ((VAR & 2) && kt > 1) return;
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(p.a + offs[region][i] + kt * 64, dst + i * 1024);
  };
  auto issue_b = [&](int kt, int region) {
    unsigned char* dst = my + (kt & 1) * KTB + region * PIECE;
    const int h = region == R_B1;
    if ((VAR & 2) && kt > 0) return;
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(bok[h][i] ? p.b + offs[region][i] + kt * 64 : zsrc, dst + i * 1024);
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets inside a piece: local row = wave part + tile * 16 + n, logical slot = ks * 4 + g
  const int sw = n & 7;
  const int so0 = ((0 + g) ^ sw) * 16, so1 = ((4 + g) ^ sw) * 16;
  const int a_row = (wr * 64 + n) * 128;                   // + (i & 3) * 16 * 128
  const int b_row = (wc * 32 + n) * 128;                   // + (j & 1) * 16 * 128

  h16x8 af[4][2], bfr[2][2];
  if (VAR & 4) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { af[i][0] = af[i][1] = h16x8{}; }
#pragma unroll
    for (int j = 0; j < 2; ++j) { bfr[j][0] = bfr[j][1] = h16x8{}; }
  }
  auto load_a = [&](const unsigned char* piece) {
    if (VAR & 4) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      af[i][0] = *reinterpret_cast<const h16x8*>(piece + a_row + i * 2048 + so0);
      af[i][1] = *reinterpret_cast<const h16x8*>(piece + a_row + i * 2048 + so1);
    }
  };
  auto load_b = [&](const unsigned char* piece) {
    if (VAR & 4) return;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bfr[j][0] = *reinterpret_cast<const h16x8*>(piece + b_row + j * 2048 + so0);
      bfr[j][1] = *reinterpret_cast<const h16x8*>(piece + b_row + j * 2048 + so1);
    }
  };

  // ---- prologue: pieces 0..4 = a0, b0, b1, a1 of K tile 0 and a0 of K tile 1 ----
  issue_a(0, R_A0);
  issue_b(0, R_B0);
  issue_b(0, R_B1);
  issue_a(0, R_A1);
  issue_a(1, R_A0);

  // phase g = 4 kt + ph: wait until pieces <= g + 1 have landed (WAIT instructions may remain in flight), meet, read this
  // phase's piece(s), issue piece g + 5 (guarded by EXISTS), 16 MFMAs
#define PHASE(WAIT, READS, ISSUE, QA, QB)              \
  {                                                     \
    wait_vmcnt<WAIT>();                                 \
    __builtin_amdgcn_s_barrier();                       \
    asm volatile("" ::: "memory");                      \
    READS;                                              \
    ISSUE;                                              \
    __builtin_amdgcn_s_setprio(1);                      \
    if (!(VAR & 1)) mma_quadrant<QA, QB>(acc, af, bfr); \
    else keep_frags(af, bfr);                           \
    __builtin_amdgcn_s_setprio(0);                      \
    asm volatile("" ::: "memory");                      \
  }

  int kt = 0;
  for (; kt < nt - 1; ++kt) {
    const unsigned char* buf = smem + (kt & 1) * KTB;
    const bool more = kt + 2 < nt;                          // K tile kt + 2 exists
    PHASE(6, { load_b(buf + R_B0 * PIECE); load_a(buf + R_A0 * PIECE); }, issue_b(kt + 1, R_B0), 0, 0)
    PHASE(6, load_b(buf + R_B1 * PIECE), issue_b(kt + 1, R_B1), 0, 1)
    PHASE(6, load_a(buf + R_A1 * PIECE), issue_a(kt + 1, R_A1), 1, 1)
    if (more) {
      PHASE(6, load_b(buf + R_B0 * PIECE), issue_a(kt + 2, R_A0), 1, 0)
    } else {
      PHASE(6, load_b(buf + R_B0 * PIECE), {}, 1, 0)
    }
  }
  {   // last K tile: nothing left to issue, the queue drains
    const unsigned char* buf = smem + (kt & 1) * KTB;
    PHASE(4, { load_b(buf + R_B0 * PIECE); load_a(buf + R_A0 * PIECE); }, {}, 0, 0)
    PHASE(2, load_b(buf + R_B1 * PIECE), {}, 0, 1)
    PHASE(0, load_a(buf + R_A1 * PIECE), {}, 1, 1)
    PHASE(0, load_b(buf + R_B0 * PIECE), {}, 1, 0)
  }
#undef PHASE

  if ((VAR & 8) && p.M > 0) {
    if (acc[0][0][0] == 12345.678f) p.out[0] = 1;    // data-dependent, never true: the accumulators stay live
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    continue;
  }
  // ---- epilogue.  A lane holds channels g*4..g*4+3 of token n of each 16 x 16 tile: stored from there, every store
  //      instruction would write 16 rows x 32 bytes, and the 100 MB of a [32768, 1536] output took longer than the whole
  //      reduction (measured: 114 us with, 49 us without the stores).  So the wave's 64 x 128 tile goes through LDS once
  //      (rows padded to 272 bytes: conflict-free 8-byte writes) and leaves as 16 bytes per lane, 256 contiguous bytes
  //      per row; bias / relu / dropout are applied in fp32 on the way in, gate / add with 16-byte loads on the way out.
  unsigned key = 0;
  if (p.thr) key = mix32((p.seed_dev ? *p.seed_dev : 0u) * 0x9E3779B1u + p.site * 0x85EBCA77u + 0x165667B1u);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                            // every wave is done with the operand buffers
  asm volatile("" ::: "memory");
  unsigned char* ep = smem + wave * EP_WAVE;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long rbase = (long)(pb * 256 + wc * 64 + j * 16 + n) * p.NO;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int co = yi * 256 + wr * 128 + i * 16 + g * 4;
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) bv = *reinterpret_cast<const f32x4*>(bias_l + wr * 128 + i * 16 + g * 4);
      h16_t outv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[i][j][r] + bv[r];
        if (p.relu) v = fmaxf(v, 0.f);
        if (p.thr) {
          const unsigned long idx = (unsigned long)(rbase + co + r);
          const unsigned hsh = mix32((unsigned)idx ^ key ^ (unsigned)(idx >> 32) * 0xC2B2AE35u);
          v = hsh >= p.thr ? v * p.keep : 0.f;
        }
        outv[r] = f2h(v);
      }
      *reinterpret_cast<uint2*>(ep + (j * 16 + n) * EP_PITCH + (i * 16 + g * 4) * 2) = *reinterpret_cast<uint2*>(outv);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the wave reads back what its own lanes wrote
  __builtin_amdgcn_wave_barrier();
  const int rrow = lane >> 4, c16 = lane & 15;
#pragma unroll 4
  for (int r = 0; r < 16; ++r) {
    const int trow = r * 4 + rrow;
    const int row = pb * 256 + wc * 64 + trow;
    if (row >= p.M) continue;
    // VAR bit 4 (measurement): every token tile writes the rows of tile 0 -- the output stays L2-resident
    const long off = (long)((VAR & 16) ? (row & 255) : row) * p.NO + yi * 256 + wr * 128 + c16 * 8;
    uint4 v = *reinterpret_cast<const uint4*>(ep + trow * EP_PITCH + c16 * 16);
    if (p.gate || p.add) {
      uint4 gv = make_uint4(0, 0, 0, 0), av = make_uint4(0, 0, 0, 0);
      if (p.gate) gv = *reinterpret_cast<const uint4*>(p.gate + off);
      if (p.add) av = *reinterpret_cast<const uint4*>(p.add + off);
      h16_t* vp = reinterpret_cast<h16_t*>(&v);
      const h16_t* gp = reinterpret_cast<const h16_t*>(&gv);
      const h16_t* ap = reinterpret_cast<const h16_t*>(&av);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float f = h2f(vp[e]);
        if (p.gate) f = h2f(gp[e]) > 0.f ? f * p.gate_pos : 0.f;
        if (p.add) f += h2f(ap[e]);
        vp[e] = f2h(f);
      }
    }
    *reinterpret_cast<uint4*>(p.out + off) = v;
  }
  // the staging rows are read: the next tile's prologue may overwrite them.  Raw barrier, no vmcnt wait -- the stores
  // stay in flight (the first counted wait of the next tile retires them, under its own DMA)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  }   // tile loop
}

// ---------------------------------------------------------------------------------------------------------------------
// gemm256_pipe (round 6): the same 256 x 256 x 64 tile and LDS-DMA ring, with the tile boundary taken out of the critical path.
//
// What the ablations of gemm256_nt say at [32768, 1536, 512] (profiles/r06_gemm256_ablation.txt): 74 us in all, 53 us
// without the output stores, and the stores cost the same 21 us when every token tile writes the SAME 256 rows (an
// L2-resident output): the epilogue is bound by its own issue -- the LDS round trip of the tile and ~10-14 B/clk/CU of
// global-store issue -- not by HBM, and nothing else runs on the CU while it lasts (one block per CU; the K = 512 layers
// have only 8 K tiles per output tile).  So here:
//   * the DMA stream never drains at a tile boundary: pieces are issued five ahead across tiles (the issuing side walks
//     its own (tile, K tile, piece) cursor), the next tile's first pieces land while the current tile finishes;
//   * the epilogue is cut into the four 64-channel x 32-token QUADRANTS of a wave's accumulators and each quadrant leaves
//     in the phase after its last MFMA, next to the following quadrant's MFMAs: phases 1, 2, 3 of the last K tile carry
//     quadrants (0,0), (0,1), (1,1), phase 0 of the NEXT tile's first K tile carries (1,0).  A quadrant goes through a
//     wave-private 2 KiB staging area (two halves of 16 tokens x 128 bytes, 16-byte-chunk swizzle) behind the ring -- no
//     block barrier, the ring is never touched -- and leaves as 16 bytes per lane, whole 128-byte row segments;
//   * the accumulators of a finished quadrant are zeroed there, so the K loop has no first-tile variant.
// gfx9 counts loads AND stores in vmcnt and retires them in order, so the counted waits of the ring have to know about the
// stores that are interleaved with the DMA: the wave keeps a small queue model in SGPRs (instructions issued in each of
// the last four phases, split into "before / after that phase's piece") and waits for exactly what the next fragment
// reads need: everything up to the piece issued four phases ago.
// Epilogues with a gate / residual operand (backward-data of linear2 / of the block inputs) stay on gemm256_nt.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int STG_WAVE = 2048;                                  // staging bytes per wave: 16 token rows x 128 bytes
constexpr int LDS_PIPE = 2 * KTB + 8 * STG_WAVE + 256 * 4;      // ring | staging | bias   (148480 bytes)

// s_waitcnt vmcnt(n) for a wave-uniform n (rounded down to an even count: waiting for one more is always safe)
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {
  switch (n >> 1) {
    case 0: wait_vmcnt<0>(); break;
    case 1: wait_vmcnt<2>(); break;
    case 2: wait_vmcnt<4>(); break;
    case 3: wait_vmcnt<6>(); break;
    case 4: wait_vmcnt<8>(); break;
    case 5: wait_vmcnt<10>(); break;
    case 6: wait_vmcnt<12>(); break;
    case 7: wait_vmcnt<14>(); break;
    case 8: wait_vmcnt<16>(); break;
    case 9: wait_vmcnt<18>(); break;
    case 10: wait_vmcnt<20>(); break;
    case 11: wait_vmcnt<22>(); break;
    case 12: wait_vmcnt<24>(); break;
    default: wait_vmcnt<26>(); break;
  }
}

template <bool DROP>
__global__ __launch_bounds__(512, 2) void gemm256_pipe(G256 p, int g_store_nt) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int wr = wave >> 2, wc = wave & 3;
  const int nt = p.K >> 6;
  const int stride = gridDim.x;
  const int nvirt = 8 * ((p.P + 7) / 8) * p.Y;

  // tile list of this block: lin = blockIdx.x + i * gridDim.x, virtual tiles beyond the token range skipped
  auto valid_from = [&](int lin) {
    while (lin < nvirt && (lin & 7) + 8 * ((lin >> 3) / p.Y) >= p.P) lin += stride;
    return lin;
  };
  auto tile_yi = [&](int lin) { return (lin >> 3) % p.Y; };
  auto tile_pb = [&](int lin) { return (lin & 7) + 8 * ((lin >> 3) / p.Y); };

  // ---- per-lane DMA source offsets WITHIN a tile (BYTES, 32 bit); the tile's base is a wave-uniform pointer, so the DMA
  //      takes the scalar-base + 32-bit-offset form and an issue costs one VALU add per instruction ----
  const int rsub = lane >> 3, pslot = lane & 7;
  // instruction i and half h move a lane's row by 8 / by 64 (a) or 32 (b) rows -- the k-slot swizzle only looks at the low
  // three row bits -- so ONE byte offset per operand is kept and the rest is a wave-uniform displacement of the base
  const int brl0 = ((wave * 16) >> 5) * 64 + ((wave * 16) & 31) + rsub;   // tile-local token row of the b pieces' instruction 0 (h = 0)
  unsigned aoff0, boff0;
  {
    const int lr = wave * 16 + rsub;
    const int c = pslot ^ (lr & 7);
    aoff0 = 2u * ((unsigned)((lr >> 6) * 128 + (lr & 63)) * (unsigned)p.K + c * 8);
    boff0 = 2u * ((unsigned)brl0 * (unsigned)p.K + c * 8);
  }
  const unsigned rowb = 2u * (unsigned)p.K;                 // bytes per operand row
  unsigned char* my = smem + wave * 2048;
  float* bias_l = reinterpret_cast<float*>(smem + 2 * KTB + 8 * STG_WAVE);
  unsigned char* stg = smem + 2 * KTB + wave * STG_WAVE;

  // ---- issuing side: cursor over (tile, K tile, piece); piece order a0, b0, b1, a1 ----
  int lin_i = valid_from(blockIdx.x);
  int kt_i = 0, kg_i = 0;
  const unsigned char* abase_i = reinterpret_cast<const unsigned char*>(p.a);
  const unsigned char* bbase_i = reinterpret_cast<const unsigned char*>(p.b);
  int pb_i = 0;
  bool full_i = true;                                       // every token row of the issue tile exists (the usual case)
  auto set_issue_tile = [&]() {
    if (lin_i < nvirt) {
      pb_i = tile_pb(lin_i);
      full_i = pb_i * 256 + 256 <= p.M;
      abase_i = reinterpret_cast<const unsigned char*>(p.a + (long)tile_yi(lin_i) * 256 * p.K);
      bbase_i = reinterpret_cast<const unsigned char*>(p.b + (long)pb_i * 256 * p.K);
    }
  };
  set_issue_tile();
  // issues piece (region R of the cursor's K tile): two DMA instructions per wave; returns how many went out (0: no tile
  // left).  The region of a phase is static -- phase ph issues region (ph + 1) & 3: b0, b1, a1, a0 of the NEXT K tile --
  // so the cursor advances behind region 3 (a1).
  auto issue_piece = [&](auto r_tag) -> int {
    constexpr int R = decltype(r_tag)::value;
    if (lin_i >= nvirt) return 0;
    unsigned char* dst = my + (kg_i & 1) * KTB + R * PIECE;
    const unsigned kob = (unsigned)kt_i * 128u;
    if constexpr (R == R_A0 || R == R_A1) {
      constexpr int h = R == R_A1;
#pragma unroll
      for (int i = 0; i < 2; ++i) glds16(abase_i + (size_t)(aoff0 + ((h * 64 + i * 8) * rowb + kob)), dst + i * 1024);
    } else {
      constexpr int h = R == R_B1;
      if (full_i) {
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(bbase_i + (size_t)(boff0 + ((h * 32 + i * 8) * rowb + kob)), dst + i * 1024);
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) {           // ragged last token tile (rare): rows beyond M read the zero page
          const bool ok = pb_i * 256 + brl0 + 8 * i + h * 32 < p.M;
          const void* zsrc = reinterpret_cast<const h16_t*>(g256_zero_page) + (lane & 7) * 8;
          glds16(ok ? reinterpret_cast<const void*>(bbase_i + (size_t)(boff0 + ((h * 32 + i * 8) * rowb + kob))) : zsrc,
                 dst + i * 1024);
        }
      }
    }
    if constexpr (R == R_A1) {
      ++kg_i;
      if (++kt_i == nt) {
        kt_i = 0;
        lin_i = valid_from(lin_i + stride);
        set_issue_tile();
      }
    }
    return 2;
  };
  auto issue_bias = [&](int yi) -> int {          // the tile's 256 bias values (every wave loads a quarter; 4-7 repeat 0-3)
    if (!p.bias) return 0;
    __builtin_amdgcn_global_load_lds((gptr_t)(p.bias + yi * 256 + (wave & 3) * 64 + lane), (lptr_t)(bias_l + (wave & 3) * 64), 4,
                                     0, 0);
    return 1;
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int sw = n & 7;
  const int so0 = ((0 + g) ^ sw) * 16, so1 = ((4 + g) ^ sw) * 16;
  const int a_row = (wr * 64 + n) * 128;
  const int b_row = (wc * 32 + n) * 128;
  h16x8 af[4][2], bfr[2][2];
  auto load_a = [&](const unsigned char* piece) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      af[i][0] = *reinterpret_cast<const h16x8*>(piece + a_row + i * 2048 + so0);
      af[i][1] = *reinterpret_cast<const h16x8*>(piece + a_row + i * 2048 + so1);
    }
  };
  auto load_b = [&](const unsigned char* piece) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bfr[j][0] = *reinterpret_cast<const h16x8*>(piece + b_row + j * 2048 + so0);
      bfr[j][1] = *reinterpret_cast<const h16x8*>(piece + b_row + j * 2048 + so1);
    }
  };

  // ---- epilogue of one quadrant (QA, QB) of the tile (eyi, epb) ----
  unsigned key = 0;
  if (DROP) key = mix32((p.seed_dev ? *p.seed_dev : 0u) * 0x9E3779B1u + p.site * 0x85EBCA77u + 0x165667B1u);
  const float lo = p.relu ? 0.f : -INFINITY;
  const unsigned drop_lane = (unsigned)n * (unsigned)p.NO + (unsigned)(g * 4);     // the lane's part of the flat output index
  // staging: row = token (n), 128 bytes = the quadrant's 64 channels; 16-byte chunk swizzle chunk ^= (row >> 1) & 7
  const int st_w = n * 128 + g * 8;                         // + ((2 i + (g >> 1)) ^ (n >> 1)) ... written out below
  const int rrow = lane >> 3, rcc = lane & 7;               // read-back: rows rrow, rrow + 8; chunk rcc
  int eyi = 0, epb = 0;
  // half J (N-tile 2 QB + J) of quadrant (QA, QB): bias, relu, dropout, pack -> 4 x 8 bytes per lane
  auto pack_half = [&](auto qa_tag, auto qb_tag, auto j_tag, uint2 (&pk)[4]) {
    constexpr int QA = decltype(qa_tag)::value, QB = decltype(qb_tag)::value, J = decltype(j_tag)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(bias_l + wr * 128 + (QA * 4 + i) * 16 + g * 4);
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = fmaxf(acc[QA * 4 + i][QB * 2 + J][r] + bv[r], lo);
        if (DROP) {
          // flat output index of gemm256_nt's hash, 32 bit (the launcher sends M x NO >= 2^32 to the round-3 kernel)
          const unsigned ubase = (unsigned)(epb * 256 + wc * 64 + (QB * 2 + J) * 16) * (unsigned)p.NO +
                                 (unsigned)(eyi * 256 + wr * 128 + (QA * 4 + i) * 16 + r);
          const unsigned hsh = mix32((ubase + drop_lane) ^ key);
          v[r] = hsh >= p.thr ? v[r] * p.keep : 0.f;
        }
      }
      pk[i] = make_uint2(f2h_pack(v[0], v[1]), f2h_pack(v[2], v[3]));
    }
  };
  auto stage_write = [&](const uint2 (&pk)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int chunk = ((2 * i + (g >> 1)) ^ (n >> 1)) & 7;
      *reinterpret_cast<uint2*>(stg + n * 128 + chunk * 16 + (g & 1) * 8) = pk[i];
    }
  };
  struct Rd2 { uint4 a, b; };
  const int rd_off0 = rrow * 128 + ((rcc ^ (rrow >> 1)) & 7) * 16;
  const int rd_off1 = (rrow + 8) * 128 + ((rcc ^ ((rrow + 8) >> 1)) & 7) * 16;
  auto stage_read = [&]() -> Rd2 {
    Rd2 r;
    r.a = *reinterpret_cast<const uint4*>(stg + rd_off0);
    r.b = *reinterpret_cast<const uint4*>(stg + rd_off1);
    return r;
  };
  // a lane's 16 bytes go to token row rrow (+ 8) of the half, chunk rcc: ONE 32-bit byte offset per lane, everything else
  // (tile, wave, quadrant, half) is a wave-uniform displacement of the base -> scalar-base stores, no per-row 64-bit products
  const unsigned st_lane = 2u * ((unsigned)rrow * (unsigned)p.NO + (unsigned)rcc * 8u);
  auto store_half = [&](auto qa_tag, auto qb_tag, auto j_tag, const Rd2& rd) -> int {
    constexpr int QA = decltype(qa_tag)::value, QB = decltype(qb_tag)::value, J = decltype(j_tag)::value;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int trow0 = epb * 256 + wc * 64 + (QB * 2 + J) * 16;                     // first token row of the half (uniform)
    unsigned char* base = reinterpret_cast<unsigned char*>(p.out + (long)trow0 * p.NO + eyi * 256 + wr * 128 + QA * 64);
    const u32x4 va = {rd.a.x, rd.a.y, rd.a.z, rd.a.w}, vb = {rd.b.x, rd.b.y, rd.b.z, rd.b.w};
    u32x4* d0 = reinterpret_cast<u32x4*>(base + (size_t)st_lane);
    u32x4* d1 = reinterpret_cast<u32x4*>(base + (size_t)(16u * (unsigned)p.NO) + (size_t)st_lane);
    const bool ok0 = trow0 + 16 <= p.M || trow0 + rrow < p.M, ok1 = trow0 + 16 <= p.M || trow0 + rrow + 8 < p.M;
    if (g_store_nt) {       // measurement switch (EVT_GEMM256_NT=1): streaming stores that do not allocate in the L2
      if (ok0) __builtin_nontemporal_store(va, d0);
      if (ok1) __builtin_nontemporal_store(vb, d1);
    } else {
      if (ok0) *d0 = va;
      if (ok1) *d1 = vb;
    }
    return 2;
  };
  auto zero_quadrant = [&](auto qa_tag, auto qb_tag) {
    constexpr int QA = decltype(qa_tag)::value, QB = decltype(qb_tag)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[QA * 4 + i][QB * 2 + j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  // ---- queue model: VMEM instructions of the last four phases (1 = previous); c = all, post = those after the piece ----
  int c1 = 2, c2 = 2, c3 = 2, post1 = 0, post2 = 0, post3 = 0, post4 = 0;

  // ---- prologue: bias of the first tile, pieces 0 .. 4 (virtual phases -5 .. -1) ----
  int lin_c = valid_from(blockIdx.x);
  if (lin_c >= nvirt) return;
  if (!p.bias && tid < 256) bias_l[tid] = 0.f;
  issue_bias(tile_yi(lin_c));
  using R0 = std::integral_constant<int, R_A0>;
  using R1 = std::integral_constant<int, R_B0>;
  using R2 = std::integral_constant<int, R_B1>;
  using R3 = std::integral_constant<int, R_A1>;
  issue_piece(R0{}); issue_piece(R1{}); issue_piece(R2{}); issue_piece(R3{}); issue_piece(R0{});

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

#define MMA_HALF(QA, QB, KS)                                                                                       \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                       \
      acc[QA * 4 + i][QB * 2 + j] = EVT_MFMA_16x16x32(af[i][KS], bfr[j][KS], acc[QA * 4 + i][QB * 2 + j], 0, 0, 0);

  // one phase: READS = this phase's fragment reads, (QA, QB) = its MFMA quadrant, EP = an epilogue rides along:
  // quadrant (EA, EB) of tile (eyi, epb); BIAS_YI >= 0: this phase also fetches the bias of the compute tile
#define PHASE(READS, RISSUE, QA, QB, EP, EA, EB, BIAS_YI)                                                                    \
  {                                                                                                                  \
    wait_vmcnt_dyn(post4 + c3 + c2 + c1);                                                                            \
    __builtin_amdgcn_s_barrier();                                                                                    \
    asm volatile("" ::: "memory");                                                                                   \
    READS;                                                                                                           \
    int pre_now = 0, post_now = 0;                                                                                   \
    if ((BIAS_YI) >= 0) pre_now = issue_bias(BIAS_YI);                                                               \
    const int dp_now = issue_piece(RISSUE{});                                                                               \
    uint2 pk[4];                                                                                                     \
    Rd2 rd;                                                                                                          \
    if (EP) {                                                                                                        \
      pack_half(I##EA{}, I##EB{}, I0{}, pk);                                                                         \
      stage_write(pk);                                                                                               \
      rd = stage_read();                                                                                             \
    }                                                                                                                \
    __builtin_amdgcn_s_setprio(1);                                                                                   \
    MMA_HALF(QA, QB, 0)                                                                                              \
    __builtin_amdgcn_s_setprio(0);                                                                                   \
    if (EP) {                                                                                                        \
      post_now += store_half(I##EA{}, I##EB{}, I0{}, rd);                                                            \
      pack_half(I##EA{}, I##EB{}, I1{}, pk);                                                                         \
      zero_quadrant(I##EA{}, I##EB{});                                                                               \
      stage_write(pk);                                                                                               \
      rd = stage_read();                                                                                             \
    }                                                                                                                \
    __builtin_amdgcn_s_setprio(1);                                                                                   \
    MMA_HALF(QA, QB, 1)                                                                                              \
    __builtin_amdgcn_s_setprio(0);                                                                                   \
    if (EP) post_now += store_half(I##EA{}, I##EB{}, I1{}, rd);                                                      \
    post4 = post3; post3 = post2; post2 = post1; post1 = post_now;                                                   \
    c3 = c2; c2 = c1; c1 = pre_now + dp_now + post_now;                                                              \
    asm volatile("" ::: "memory");                                                                                   \
  }

  // one K tile: EP0 = phase 0 carries quadrant (1,0) of the PREVIOUS tile and phase 1 fetches this tile's bias;
  // EP123 = the last K tile: phases 1, 2, 3 carry quadrants (0,0), (0,1), (1,1) of this tile
#define KTILE(EP0, EP123, YI)                                                                                        \
  {                                                                                                                  \
    const unsigned char* buf = smem + (kg_c & 1) * KTB;                                                              \
    PHASE({ load_b(buf + R_B0 * PIECE); load_a(buf + R_A0 * PIECE); }, R1, 0, 0, EP0, 1, 0, -1)                          \
    if (EP0) { eyi = tile_yi(lin_c); epb = tile_pb(lin_c); }                                                         \
    PHASE(load_b(buf + R_B1 * PIECE), R2, 0, 1, EP123, 0, 0, (EP0) ? (YI) : -1)                                          \
    PHASE(load_a(buf + R_A1 * PIECE), R3, 1, 1, EP123, 0, 1, -1)                                                         \
    PHASE(load_b(buf + R_B0 * PIECE), R0, 1, 0, EP123, 1, 1, -1)                                                         \
    ++kg_c;                                                                                                          \
  }

  int kg_c = 0;
  bool has_prev = false;
  eyi = tile_yi(lin_c);
  epb = tile_pb(lin_c);
  while (lin_c < nvirt) {
    const int yi_c = tile_yi(lin_c);
    if (has_prev) KTILE(1, 0, yi_c) else KTILE(0, 0, yi_c)
    for (int kt = 1; kt < nt - 1; ++kt) KTILE(0, 0, yi_c)
    KTILE(0, 1, yi_c)
    has_prev = true;
    lin_c = valid_from(lin_c + stride);
  }
#undef KTILE
#undef PHASE
#undef MMA_HALF
  // ---- the last tile's quadrant (1,0): nothing left to multiply ----
  {
    uint2 pk[4];
    pack_half(I1{}, I0{}, I0{}, pk);
    stage_write(pk);
    Rd2 rd = stage_read();
    store_half(I1{}, I0{}, I0{}, rd);
    pack_half(I1{}, I0{}, I1{}, pk);
    stage_write(pk);
    rd = stage_read();
    store_half(I1{}, I0{}, I1{}, rd);
  }
}

int g_variant = 0;     // measurement switch (evt_debug_gemm256_variant); 0 = the product kernel

bool eligible(const evt_gemm_params* g, int kred, int nout) {
  if (g->dtype != EVT_DT_HALF) return false;
  if (nout % 256 || kred % 64 || kred < 128) return false;
  if ((long)g->M * kred >= (1L << 31) || (long)nout * kred >= (1L << 31)) return false;
  if (g->M < 2048) return false;                             // few token tiles: the 128 / 64 tiles fill the chip better
  return getenv("EVT_NO_GEMM256") == nullptr;                // A/B switch (measurements, fused-vs-composed tests)
}

int launch(const evt_gemm_params* g, const void* a, const void* b, int kred, int nout, const float* bias, int relu,
           const evt_gemm_epilogue* e, void* out, hipStream_t st) {
  G256 p{};
  p.a = (const h16_t*)a; p.b = (const h16_t*)b; p.out = (h16_t*)out; p.bias = bias;
  p.M = g->M; p.NO = nout; p.K = kred; p.relu = relu;
  p.keep = 1.f; p.gate_pos = 1.f;
  if (e) {
    if (e->dropout_p < 0.f || e->dropout_p >= 1.f) return EVT_EINVAL;
    p.gate = (const h16_t*)e->gate; p.add = (const h16_t*)e->add; p.gate_pos = e->gate_pos;
    p.seed_dev = e->seed_dev; p.site = e->site;
    if (e->dropout_p > 0.f) {
      p.thr = (unsigned)fminf(e->dropout_p * 4294967296.f, 4294967040.f);
      p.keep = 1.f / (1.f - e->dropout_p);
    }
  }
  p.Y = nout / 256;
  p.P = (g->M + 255) / 256;
  const size_t lds = LDS_TOTAL;
  const int nvirt = 8 * ((p.P + 7) / 8) * p.Y;
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return EVT_ELAUNCH;
    ncu = prop.multiProcessorCount / 8 * 8;
    if (ncu <= 0) ncu = 8;
  }
  const dim3 grid(nvirt < ncu ? nvirt : ncu);
  // the pipelined kernel takes every launch without a gate / residual operand (EVT_GEMM256_PIPE=0: the round-3 kernel)
  static const bool pipe_on = !(getenv("EVT_GEMM256_PIPE") && atoi(getenv("EVT_GEMM256_PIPE")) == 0);
  // (one long-K tile per block -- [32768, 512, 2048]: 256 tiles, 32 K tiles each -- has no boundary to hide and runs 5 % faster
  //  on the round-3 loop: 68 against 72 us)
  static const int pipe_force = getenv("EVT_GEMM256_PIPE") ? atoi(getenv("EVT_GEMM256_PIPE")) : 1;
  const bool pipe_shape = nvirt > (int)grid.x || (p.K >> 6) <= 12 || pipe_force == 2;
  if (pipe_on && pipe_shape && g_variant == 0 && !p.gate && !p.add && (long)p.M * p.NO < (1L << 32)) {
    evt_set_last_tag("gemm256_pipe<bf16, 256, 256, 64>");
    static bool attr_p[2] = {false, false};
    const int d = p.thr ? 1 : 0;
    const void* fn = d ? reinterpret_cast<const void*>(&gemm256_pipe<true>) : reinterpret_cast<const void*>(&gemm256_pipe<false>);
    if (!attr_p[d]) {
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_PIPE) != hipSuccess) return EVT_ELAUNCH;
      attr_p[d] = true;
    }
    static const int nt_st = getenv("EVT_GEMM256_NT") ? atoi(getenv("EVT_GEMM256_NT")) : 0;
    if (d) hipLaunchKernelGGL(gemm256_pipe<true>, grid, dim3(512), LDS_PIPE, st, p, nt_st);
    else hipLaunchKernelGGL(gemm256_pipe<false>, grid, dim3(512), LDS_PIPE, st, p, nt_st);
    return evt_check_launch();
  }
  evt_set_last_tag("gemm256_nt<bf16, 256, 256, 64>");
#define G256_LAUNCH(V)                                                                                                  \
  {                                                                                                                     \
    static bool attr = false;                                                                                           \
    if (!attr) {                                                                                                        \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_nt<V>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)lds) != hipSuccess)                                                                  \
        return EVT_ELAUNCH;                                                                                             \
      attr = true;                                                                                                      \
    }                                                                                                                   \
    hipLaunchKernelGGL(gemm256_nt<V>, grid, dim3(512), lds, st, p);                                                     \
  }
  switch (g_variant) {
    case 0: G256_LAUNCH(0) break;
    case 1: G256_LAUNCH(1) break;
    case 2: G256_LAUNCH(2) break;
    case 4: G256_LAUNCH(4) break;
    case 5: G256_LAUNCH(5) break;
    case 6: G256_LAUNCH(6) break;
    case 8: G256_LAUNCH(8) break;
    case 9: G256_LAUNCH(9) break;
    case 16: G256_LAUNCH(16) break;
    case 17: G256_LAUNCH(17) break;
    default: return EVT_EINVAL;
  }
#undef G256_LAUNCH
  return evt_check_launch();
}

}  // namespace

extern "C" {

void evt_debug_gemm256_variant(int32_t v) { g_variant = v; }

int32_t evt_gemm_bf16_fused_supported(const evt_gemm_params* g, int32_t backward_data) {
  if (!g || g->M <= 0 || g->N <= 0 || g->K <= 0) return 0;
  return backward_data ? eligible(g, g->N, g->K) : eligible(g, g->K, g->N);
}

int evt_gemm_bf16_fwd_ex(const evt_gemm_params* g, const void* x, const void* w_reg, const void* w_alt, const float* bias,
                         const evt_gemm_epilogue* epi, void* y, void* stream) {
  if (!g || !x || !w_reg || !y) return EVT_EINVAL;
  if (!eligible(g, g->K, g->N)) return EVT_ENOTSUP;
  (void)w_alt;
  return launch(g, w_reg, x, g->K, g->N, bias, g->relu, epi, y, (hipStream_t)stream);
}

int evt_gemm_bf16_bwd_data_ex(const evt_gemm_params* g, const void* dy, const void* w_reg, const void* w_alt,
                              const evt_gemm_epilogue* epi, void* dx, void* stream) {
  if (!g || !dy || !w_alt || !dx) return EVT_EINVAL;
  if (!eligible(g, g->N, g->K)) return EVT_ENOTSUP;
  (void)w_reg;
  return launch(g, w_alt, dy, g->N, g->K, nullptr, 0, epi, dx, (hipStream_t)stream);
}

}  // extern "C"
