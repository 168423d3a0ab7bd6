// conv_deep: GEMM-grade implicit-GEMM convolution for the wide bf16 layers (gfx950).
//
// Reference call sites: DiscriminatorP's 128->512, 512->1024 (k5 s3) and 1024->1024 (k5 s1) Conv2d((k,1)) layers,
// src/easevoice/module/models.py:481-536, forward and backward-data; the same descriptor (conv_p.h) as conv_igemm.
//
// GEMM view: M = output channels, N = FLAT positions (seq, q) of all sequences (a 128-position tile may span several
// of DiscriminatorP's short sequences), K = (tap, K-side channel).  Block tile 128 x 128, 4 waves in 2 x 2, a wave
// owns 64 x 64 = 4 x 4 MFMA tiles (mfma_f32_16x16x32_bf16).  One K stage = one tap x 64 channels:
//   A stage  [128 out-channels][64 k]  from the prepared weight image [co][chunk32][tap][32] (two 64-byte pieces)
//   B stage  [128 positions][64 k]     row (seq, q*s_in + tap*dil + off_in), 128 contiguous bytes of a channels-last row
// Both are written by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass).  The DMA destination
// is lane-linear, so the bank-conflict swizzle is applied to the per-lane SOURCE address and undone with the same XOR
// on the fragment read: LDS[row][slot] = G[row][slot ^ (row & 7)] (16-byte slots of a 128-byte row; every
// ds_read_b128 lane group then touches 16 distinct slots).  Rows outside a sequence (conv padding, tile tail) read a
// zero page.  Two LDS stages (64 KiB): the DMA of stage s+1 is issued right after the barrier that publishes stage s
// and flies under its 32 MFMAs per wave; one barrier per stage.
#include "conv_p.h"
#include "wgrad_epi.h"
#include <cstdlib>

namespace evt_conv {
namespace {

__device__ __attribute__((aligned(256))) unsigned int g_zero_page[64];  // 256 zero bytes

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
}

constexpr int BM = 128, BN = 128, BK = 64;

// -----------------------------------------------------------------------------------------------------------------
// Epilogue of the LDS-DMA conv kernels.  An MFMA lane holds 4 consecutive output channels of ONE position: stored from
// there, an instruction writes 16 rows x 32 bytes (measured on the 256 x 256 GEMM: 100 MB of output took 65 us that way,
// 19 us as whole rows).  So the block's [TN positions][TM channels] tile is staged in LDS once -- rows padded by 16 bytes:
// conflict-free 8-byte writes -- and leaves as 16 bytes per lane, whole rows of 2*TM contiguous bytes.  bias / output
// activation are applied in fp32 on the way in; gate / residual with 16-byte loads on the way out.  The operand stages
// are dead by then (every wave passed the last stage's MFMAs before the barrier).
// -----------------------------------------------------------------------------------------------------------------
template <int TM, int TN, int MI, int NJ>
__device__ __forceinline__ void store_tile_staged(const ConvP& p, unsigned char* smem, const f32x4 (&acc)[MI][NJ], int wr,
                                                  int wc, int n, int g, int yi, int pb, int phase, int total_units) {
  constexpr int PITCH = TM * 2 + 16;
  constexpr int CPR = TM / 8;                              // 16-byte chunks per row
  constexpr int RPP = 256 / CPR;                           // rows per pass of the block
  f32x4 bv[MI];                                            // this lane's bias values: one 16-byte load per channel tile
#pragma unroll
  for (int i = 0; i < MI; ++i)
    bv[i] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + yi * TM + wr * 16 * MI + i * 16 + g * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int pl = wc * 16 * NJ + j * 16 + n;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int cl = wr * 16 * MI + i * 16 + g * 4;
      h16_t outv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[i][j][r] + bv[i][r];
        if (p.out_act == EVT_ACT_LRELU) v = lrelu_f(v, p.out_slope);
        else if (p.out_act == EVT_ACT_TANH) v = tanhf(v);
        outv[r] = f2h(v);
      }
      *reinterpret_cast<uint2*>(smem + pl * PITCH + cl * 2) = *reinterpret_cast<uint2*>(outv);
    }
  }
  __syncthreads();
  h16_t* Y = reinterpret_cast<h16_t*>(p.y);
  const h16_t* R = reinterpret_cast<const h16_t*>(p.res);
  const h16_t* G = reinterpret_cast<const h16_t*>(p.gate);
  const int tid = threadIdx.x;
  const int c16 = tid % CPR, r0 = tid / CPR;
#pragma unroll
  for (int pass = 0; pass * RPP < TN; ++pass) {
    const int pl = pass * RPP + r0;
    const int u = pb * TN + pl;
    if (pl >= TN || u >= total_units) continue;
    const int seq = u / p.Q;
    const int q = u - seq * p.Q;
    const int orow = q * p.s_out + p.off_out + phase * p.off_out_phase;
    if (orow < 0 || orow >= p.Lout) continue;
    const long off = ((long)seq * p.Lout + orow) * p.Cout + yi * TM + c16 * 8;
    uint4 v = *reinterpret_cast<const uint4*>(smem + pl * PITCH + c16 * 16);
    if (G || R) {
      uint4 gv = make_uint4(0, 0, 0, 0), rv = make_uint4(0, 0, 0, 0);
      if (G) gv = *reinterpret_cast<const uint4*>(G + off);
      if (R) rv = *reinterpret_cast<const uint4*>(R + off);
      h16_t* vp = reinterpret_cast<h16_t*>(&v);
      const h16_t* gp = reinterpret_cast<const h16_t*>(&gv);
      const h16_t* rp = reinterpret_cast<const h16_t*>(&rv);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float f = h2f(vp[e]);
        if (G) f *= (h2f(gp[e]) > 0.f ? 1.f : p.gate_slope);
        if (R) f += h2f(rp[e]);
        vp[e] = f2h(f);
      }
    }
    *reinterpret_cast<uint4*>(Y + off) = v;
  }
}


constexpr int STAGE_BYTES = (BM + BN) * BK * 2;  // 32 KiB

__global__ __launch_bounds__(256, 2) void conv_deep(ConvP p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;

  // XCD-aware decode (same convention as conv_igemm): all channel tiles of a position tile on one XCD
  const int lin = blockIdx.x;
  const int xcd = lin & 7, slot = lin >> 3;
  const int yi = slot % p.Y;
  const int pb = xcd + 8 * (slot / p.Y);
  if (pb >= p.P) return;
  const int phase = blockIdx.y;

  const h16_t* X = reinterpret_cast<const h16_t*>(p.x);
  const h16_t* W = reinterpret_cast<const h16_t*>(p.w) + (long)phase * p.w_phase_stride;
  const int total_units = p.nseq * p.Q;

  // ---- per-lane DMA sources: wave w stages rows [32w, 32w+32) of both tiles, 8 rows per instruction ----
  const int rsub = lane >> 3, pslot = lane & 7;
  long aoff[4], boff[4];
  int brow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave * 32 + i * 8 + rsub;
    const int c = pslot ^ (row & 7);                 // logical 16-byte k-slot this lane fetches
    const int co = yi * BM + row;
    aoff[i] = ((long)co * p.nchunk + (c >> 2)) * p.KHp * 32 + (c & 3) * 8;
    const int u = pb * BN + row;                      // launcher guarantees nseq * Q < 2^31
    const bool ok = u < total_units;
    const int seq = ok ? u / p.Q : 0;
    const int q = ok ? u - seq * p.Q : 0;
    brow[i] = ok ? q * p.s_in + p.off_in : -(1 << 28);
    boff[i] = ((long)seq * p.Lin + brow[i]) * p.Cin + c * 8;
  }
  const h16_t* zsrc = reinterpret_cast<const h16_t*>(g_zero_page) + pslot * 8;
  unsigned char* my_a = smem + wave * 32 * 128;              // + buf*STAGE_BYTES + i*1024
  unsigned char* my_b = smem + BM * 128 + wave * 32 * 128;

  const int nch2 = p.nchunk >> 1;
  const int nst = nch2 * p.KHp;

  auto issue = [&](int st, int buf) {
    const int ch2 = st / p.KHp, tap = st - ch2 * p.KHp;
    const long wsoff = ((long)(2 * ch2) * p.KHp + tap) * 32;
    const int rshift = tap * p.dil;
    const long xsoff = (long)rshift * p.Cin + ch2 * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(W + aoff[i] + wsoff, my_a + buf * STAGE_BYTES + i * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = (unsigned)(brow[i] + rshift) < (unsigned)p.Lin;
      const h16_t* src = ok ? X + boff[i] + xsoff : zsrc;
      glds16(src, my_b + buf * STAGE_BYTES + i * 1024);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets: row = tile_row0 + t*16 + n, logical slot c = ks*4 + g, physical slot = c ^ (row & 7);
  // (row & 7) == (n & 7) because every tile row base is a multiple of 16
  const int sw = n & 7;
  const int a_base = (wr * 64 + n) * 128;
  const int b_base = BM * 128 + (wc * 64 + n) * 128;
  const int so0 = ((0 + g) ^ sw) * 16, so1 = ((4 + g) ^ sw) * 16;

  issue(0, 0);
  for (int st = 0; st < nst; ++st) {
    const int buf = st & 1;
    __syncthreads();                       // (compiler drains vmcnt before it) stage st has landed; buf^1 is free
    if (st + 1 < nst) issue(st + 1, buf ^ 1);
    const unsigned char* sa = smem + buf * STAGE_BYTES + a_base;
    const unsigned char* sb = smem + buf * STAGE_BYTES + b_base;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int so = ks ? so1 : so0;
      h16x8 a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const h16x8*>(sa + i * 16 * 128 + so);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const h16x8*>(sb + j * 16 * 128 + so);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = EVT_MFMA_16x16x32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }

  // ---- epilogue: through LDS, whole rows (store_tile_staged) ----
  store_tile_staged<BM, BN, 4, 4>(p, smem, acc, wr, wc, n, g, yi, pb, phase, total_units);
}

// -----------------------------------------------------------------------------------------------------------------
// conv_deep32: the same 128 x 128 LDS-DMA GEMM with K stages of ONE tap x 32 channels (64-byte tile rows).  Half the LDS
// per stage (2 x 16 KiB per block) doubles the resident blocks per CU: the PMC profile of conv_deep showed a third of the
// wave cycles parked at the stage barrier waiting for the DMA -- more resident waves cover that, and K-side widths that
// are multiples of 32 but not of 64 (32, 96) become eligible.  Swizzle for 64-byte rows: slot ^= 2*((row>>3)&1).
// -----------------------------------------------------------------------------------------------------------------
constexpr int STAGE32 = (BM + BN) * 64;   // 16 KiB

__global__ __launch_bounds__(256, 4) void conv_deep32(ConvP p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  const int lin = blockIdx.x;
  const int xcd = lin & 7, slot = lin >> 3;
  const int yi = slot % p.Y;
  const int pb = xcd + 8 * (slot / p.Y);
  if (pb >= p.P) return;
  const int phase = blockIdx.y;
  const h16_t* X = reinterpret_cast<const h16_t*>(p.x);
  const h16_t* W = reinterpret_cast<const h16_t*>(p.w) + (long)phase * p.w_phase_stride;
  const int total_units = p.nseq * p.Q;

  // wave w stages rows [32w, 32w+32) of both tiles, 16 rows x 4 slots per instruction
  const int rsub = lane >> 2, pslot = lane & 3;
  long aoff[2], boff[2];
  int brow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = wave * 32 + i * 16 + rsub;
    const int c = pslot ^ (2 * ((row >> 3) & 1));
    aoff[i] = (long)(yi * BM + row) * p.nchunk * p.KHp * 32 + c * 8;
    const int u = pb * BN + row;
    const bool ok = u < total_units;
    const int seq = ok ? u / p.Q : 0;
    const int q = ok ? u - seq * p.Q : 0;
    brow[i] = ok ? q * p.s_in + p.off_in : -(1 << 28);
    boff[i] = ((long)seq * p.Lin + brow[i]) * p.Cin + c * 8;
  }
  const h16_t* zsrc = reinterpret_cast<const h16_t*>(g_zero_page) + pslot * 8;
  unsigned char* my_a = smem + wave * 32 * 64;
  unsigned char* my_b = smem + BM * 64 + wave * 32 * 64;
  const int nst = p.nchunk * p.KHp;

  auto issue = [&](int st, int buf) {
    const int ch = st / p.KHp, tap = st - ch * p.KHp;
    const long wsoff = ((long)ch * p.KHp + tap) * 32;
    const int rshift = tap * p.dil;
    const long xsoff = (long)rshift * p.Cin + ch * 32;
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(W + aoff[i] + wsoff, my_a + buf * STAGE32 + i * 1024);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool ok = (unsigned)(brow[i] + rshift) < (unsigned)p.Lin;
      glds16(ok ? X + boff[i] + xsoff : zsrc, my_b + buf * STAGE32 + i * 1024);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int so = ((g ^ (2 * ((n >> 3) & 1))) * 16);
  const int a_base = (wr * 64 + n) * 64 + so;
  const int b_base = BM * 64 + (wc * 64 + n) * 64 + so;

  issue(0, 0);
  for (int st = 0; st < nst; ++st) {
    const int buf = st & 1;
    __syncthreads();
    if (st + 1 < nst) issue(st + 1, buf ^ 1);
    const unsigned char* sa = smem + buf * STAGE32 + a_base;
    const unsigned char* sb = smem + buf * STAGE32 + b_base;
    h16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const h16x8*>(sa + i * 16 * 64);
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const h16x8*>(sb + j * 16 * 64);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = EVT_MFMA_16x16x32(a[i], b[j], acc[i][j], 0, 0, 0);
  }

  store_tile_staged<BM, BN, 4, 4>(p, smem, acc, wr, wc, n, g, yi, pb, phase, total_units);
}

// -----------------------------------------------------------------------------------------------------------------
// conv_ring<MT, NT, NS>: the same LDS-DMA implicit GEMM with a smaller block tile (32*MT x 32*NT, 2 x 2 waves of
// 16*MT x 16*NT) and an NS-deep ring of K stages for the LATENCY-bound layers: WN in/res_skip convs and the FFN convs of
// the encoders are 3200-position problems with 6..24 K stages -- with one stage in flight every stage costs a full
// HBM/L2 round trip (~2 us) for ~0.1 us of MFMA work.  Here NS-1 stages are in flight: the DMA of stage s+NS-1 is issued
// when stage s is published, waits are counted (s_waitcnt vmcnt(N), never a drain in steady state) and the block meets
// at ONE raw s_barrier per stage:
//     wait for MY pieces of stage s  ->  s_barrier (everyone's pieces landed; stage s-1 is free)  ->  issue stage s+NS-1
//     into the freed buffer  ->  fragment reads + MFMAs of stage s
// -----------------------------------------------------------------------------------------------------------------
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// wait until at most `ahead` groups of G DMA instructions are still in flight (ahead <= MAXA, wave-uniform)
template <int G, int MAXA>
__device__ __forceinline__ void wait_groups(int ahead) {
  if constexpr (MAXA <= 0) { wait_vmcnt<0>(); }
  else {
    if (ahead >= MAXA) wait_vmcnt<(MAXA * G > 63 ? 63 : MAXA * G)>();
    else wait_groups<G, MAXA - 1>(ahead);
  }
}

template <int MT, int NT, int NS>
__global__ __launch_bounds__(256, 2) void conv_ring(ConvP p) {
  constexpr int RM = 32 * MT, RN = 32 * NT;               // block tile
  constexpr int RSTAGE = (RM + RN) * 128;                  // bytes per stage
  constexpr int G = MT + NT;                               // DMA instructions per wave per stage
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;

  const int lin = blockIdx.x;
  const int xcd = lin & 7, slot = lin >> 3;
  const int yi = slot % p.Y;
  const int pb = xcd + 8 * (slot / p.Y);
  if (pb >= p.P) return;
  const int phase = blockIdx.y;

  const h16_t* X = reinterpret_cast<const h16_t*>(p.x);
  const h16_t* W = reinterpret_cast<const h16_t*>(p.w) + (long)phase * p.w_phase_stride;
  const int total_units = p.nseq * p.Q;

  // wave w stages rows [w*RM/4, (w+1)*RM/4) of A and [w*RN/4, ...) of B, 8 rows per instruction
  const int rsub = lane >> 3, pslot = lane & 7;
  long aoff[MT], boff[NT];
  int brow[NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int row = wave * (RM / 4) + i * 8 + rsub;
    const int c = pslot ^ (row & 7);
    aoff[i] = ((long)(yi * RM + row) * p.nchunk + (c >> 2)) * p.KHp * 32 + (c & 3) * 8;
  }
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int row = wave * (RN / 4) + i * 8 + rsub;
    const int c = pslot ^ (row & 7);
    const int u = pb * RN + row;
    const bool ok = u < total_units;
    const int seq = ok ? u / p.Q : 0;
    const int q = ok ? u - seq * p.Q : 0;
    brow[i] = ok ? q * p.s_in + p.off_in : -(1 << 28);
    boff[i] = ((long)seq * p.Lin + brow[i]) * p.Cin + c * 8;
  }
  const h16_t* zsrc = reinterpret_cast<const h16_t*>(g_zero_page) + pslot * 8;
  unsigned char* my_a = smem + wave * (RM / 4) * 128;
  unsigned char* my_b = smem + RM * 128 + wave * (RN / 4) * 128;
  const int nst = (p.nchunk >> 1) * p.KHp;

  auto issue = [&](int st) {
    const int buf = st % NS;
    const int ch2 = st / p.KHp, tap = st - ch2 * p.KHp;
    const long wsoff = ((long)(2 * ch2) * p.KHp + tap) * 32;
    const int rshift = tap * p.dil;
    const long xsoff = (long)rshift * p.Cin + ch2 * 64;
#pragma unroll
    for (int i = 0; i < MT; ++i) glds16(W + aoff[i] + wsoff, my_a + buf * RSTAGE + i * 1024);
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const bool ok = (unsigned)(brow[i] + rshift) < (unsigned)p.Lin;
      glds16(ok ? X + boff[i] + xsoff : zsrc, my_b + buf * RSTAGE + i * 1024);
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int sw = n & 7;
  const int a_base = (wr * 16 * MT + n) * 128;
  const int b_base = RM * 128 + (wc * 16 * NT + n) * 128;
  const int so0 = ((0 + g) ^ sw) * 16, so1 = ((4 + g) ^ sw) * 16;

#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nst) issue(s);
  for (int st = 0; st < nst; ++st) {
    // groups still allowed in flight after stage st has landed: stages st+1 .. st+NS-2 that exist
    const int ahead = min(NS - 2, nst - 1 - st);
    wait_groups<G, NS - 2>(ahead);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (st + NS - 1 < nst) issue(st + NS - 1);
    const unsigned char* sa = smem + (st % NS) * RSTAGE + a_base;
    const unsigned char* sb = smem + (st % NS) * RSTAGE + b_base;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int so = ks ? so1 : so0;
      h16x8 a[MT], b[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const h16x8*>(sa + i * 16 * 128 + so);
#pragma unroll
      for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const h16x8*>(sb + j * 16 * 128 + so);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = EVT_MFMA_16x16x32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("" ::: "memory");
  }

  store_tile_staged<RM, RN, MT, NT>(p, smem, acc, wr, wc, n, g, yi, pb, phase, total_units);
}


// -----------------------------------------------------------------------------------------------------------------
// wgrad_deep: weight gradient of the same layers on the same LDS-DMA structure.
//   dW[a][chunk][tap][cc] += sum over flat positions u = (seq, q) of A[u][a] * B[seq][q*s + tap*dil + off][chunk*32+cc]
// GEMM view: M = A channels (dy), N = (tap, B channel), K = flat positions.  Block tile: 128 A channels x (KT = 5 taps
// x 32 B channels), K stage = 64 positions; 4 waves in 2 x 2, a wave owns 64 A channels x 16 B channels x 5 taps
// (4 x 5 MFMA tiles).  Both operands are position-major in HBM (channels-last), i.e. K is the ROW index, so the MFMA
// fragments (8 consecutive positions of one channel) come from ds_read_b64_tr_b16 (LDS transpose read; inside a 16-lane
// group lane j supplies row j>>2, columns 4*(j&3).., lane i receives column i -- tools/probe_tr.hip).
// LDS stage: A tile [64 pos][128 ch] (256-byte rows) + KT tiles [64 pos][32 ch] (64-byte rows), written by LDS-DMA;
// the 16-byte-slot swizzles (applied on the DMA source, undone on the read) make every 32-lane transpose read touch
// 16 distinct slots:  A: slot ^= 2*((row&3) | ((row>>3)&1)<<2),   B: slot ^= 2*((row>>3)&1).
// Positions are split over blockIdx.y; partial tiles are accumulated into the fp32 dW image with atomics.
// -----------------------------------------------------------------------------------------------------------------
constexpr int WKT = 5;                                   // taps per block
constexpr int WPOS = 64;                                 // positions per K stage
constexpr int WA_BYTES = WPOS * 256;                     // 16 KiB
constexpr int WB_BYTES = WPOS * 64;                      // 4 KiB per tap
constexpr int WSTAGE = WA_BYTES + WKT * WB_BYTES;        // 36 KiB

// All 18 transpose reads of one K = 32 step (4 A tiles, 5 tap tiles, low + high halves) behind ONE wait; the row /
// tap / k-step displacements are instruction offsets, so the step needs five address registers.
template <int KS>
__device__ __forceinline__ void tr_load_step(const unsigned (&aa)[4], unsigned ba, h16x8 (&a)[4], h16x8 (&b)[WKT]) {
  uint2 al[4], ah[4], bl[WKT], bh[WKT];
  if constexpr (KS == 0) {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %18 offset:0\n\t"
        "ds_read_b64_tr_b16 %4, %18 offset:1024\n\t"
        "ds_read_b64_tr_b16 %1, %19 offset:0\n\t"
        "ds_read_b64_tr_b16 %5, %19 offset:1024\n\t"
        "ds_read_b64_tr_b16 %2, %20 offset:0\n\t"
        "ds_read_b64_tr_b16 %6, %20 offset:1024\n\t"
        "ds_read_b64_tr_b16 %3, %21 offset:0\n\t"
        "ds_read_b64_tr_b16 %7, %21 offset:1024\n\t"
        "ds_read_b64_tr_b16 %8, %22 offset:0\n\t"
        "ds_read_b64_tr_b16 %13, %22 offset:256\n\t"
        "ds_read_b64_tr_b16 %9, %22 offset:4096\n\t"
        "ds_read_b64_tr_b16 %14, %22 offset:4352\n\t"
        "ds_read_b64_tr_b16 %10, %22 offset:8192\n\t"
        "ds_read_b64_tr_b16 %15, %22 offset:8448\n\t"
        "ds_read_b64_tr_b16 %11, %22 offset:12288\n\t"
        "ds_read_b64_tr_b16 %16, %22 offset:12544\n\t"
        "ds_read_b64_tr_b16 %12, %22 offset:16384\n\t"
        "ds_read_b64_tr_b16 %17, %22 offset:16640\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : "=&v"(al[0]), "=&v"(al[1]), "=&v"(al[2]), "=&v"(al[3]), "=&v"(ah[0]), "=&v"(ah[1]), "=&v"(ah[2]), "=&v"(ah[3]),
          "=&v"(bl[0]), "=&v"(bl[1]), "=&v"(bl[2]), "=&v"(bl[3]), "=&v"(bl[4]), "=&v"(bh[0]), "=&v"(bh[1]), "=&v"(bh[2]),
          "=&v"(bh[3]), "=&v"(bh[4])
        : "v"(aa[0]), "v"(aa[1]), "v"(aa[2]), "v"(aa[3]), "v"(ba)
        : "memory");
  } else {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %18 offset:8192\n\t"
        "ds_read_b64_tr_b16 %4, %18 offset:9216\n\t"
        "ds_read_b64_tr_b16 %1, %19 offset:8192\n\t"
        "ds_read_b64_tr_b16 %5, %19 offset:9216\n\t"
        "ds_read_b64_tr_b16 %2, %20 offset:8192\n\t"
        "ds_read_b64_tr_b16 %6, %20 offset:9216\n\t"
        "ds_read_b64_tr_b16 %3, %21 offset:8192\n\t"
        "ds_read_b64_tr_b16 %7, %21 offset:9216\n\t"
        "ds_read_b64_tr_b16 %8, %22 offset:2048\n\t"
        "ds_read_b64_tr_b16 %13, %22 offset:2304\n\t"
        "ds_read_b64_tr_b16 %9, %22 offset:6144\n\t"
        "ds_read_b64_tr_b16 %14, %22 offset:6400\n\t"
        "ds_read_b64_tr_b16 %10, %22 offset:10240\n\t"
        "ds_read_b64_tr_b16 %15, %22 offset:10496\n\t"
        "ds_read_b64_tr_b16 %11, %22 offset:14336\n\t"
        "ds_read_b64_tr_b16 %16, %22 offset:14592\n\t"
        "ds_read_b64_tr_b16 %12, %22 offset:18432\n\t"
        "ds_read_b64_tr_b16 %17, %22 offset:18688\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : "=&v"(al[0]), "=&v"(al[1]), "=&v"(al[2]), "=&v"(al[3]), "=&v"(ah[0]), "=&v"(ah[1]), "=&v"(ah[2]), "=&v"(ah[3]),
          "=&v"(bl[0]), "=&v"(bl[1]), "=&v"(bl[2]), "=&v"(bl[3]), "=&v"(bl[4]), "=&v"(bh[0]), "=&v"(bh[1]), "=&v"(bh[2]),
          "=&v"(bh[3]), "=&v"(bh[4])
        : "v"(aa[0]), "v"(aa[1]), "v"(aa[2]), "v"(aa[3]), "v"(ba)
        : "memory");
  }
  union { uint4 u; h16x8 v; } r;
#pragma unroll
  for (int i = 0; i < 4; ++i) { r.u = make_uint4(al[i].x, al[i].y, ah[i].x, ah[i].y); a[i] = r.v; }
#pragma unroll
  for (int t = 0; t < WKT; ++t) { r.u = make_uint4(bl[t].x, bl[t].y, bh[t].x, bh[t].y); b[t] = r.v; }
}

__global__ __launch_bounds__(256, 2) void wgrad_deep(WgP p, int stages_per_split) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j16 = lane & 15, g8 = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;

  // Block -> (dy tile, x chunk, tap group).  A block reads the dy columns of its A tile and the x columns of its chunk for
  // its split's positions; consecutive workgroup ids go round-robin over the 8 XCDs, so with the plain decode (chunk
  // fastest) every L2 pulls ALL dy tiles and 1/8 of the x chunks: (8 A + B) / (A + B) = 4.5 x the operand bytes at
  // 1024 -> 1024 (PMC round 3: FETCH 105 MB per launch against 27.5 MB of operands).  xcd_order: the tile grid is cut into
  // 2 (dy-tile halves) x 4 (chunk quarters) rectangles, one per XCD -- an L2 sees half of dy and a quarter of x:
  // 8 x (A/2 + B/4) = 3 x.  (A split of the positions over the XCDs would give 1 x, but the split count is bounded by
  // the slabs of the deterministic reduction: 3 for a 21 MB image.)
  int bx = blockIdx.x;
  int ch, tgi, atile;
  if (p.xcd_order) {
    const int xcd = bx & 7, j = bx >> 3;
    const int na = (p.CA >> 7) >> 1, nc = p.nchunk >> 2;           // tiles per rectangle side
    const int cl = j % nc, r = j / nc;
    tgi = r % p.ntapgrp;
    const int al = r / p.ntapgrp;
    atile = (xcd & 1) * na + al;
    ch = (xcd >> 1) * nc + cl;
  } else {
    ch = bx % p.nchunk; bx /= p.nchunk;
    tgi = bx % p.ntapgrp;
    atile = bx / p.ntapgrp;
  }
  const int t0 = tgi * WKT;
  const int ntap = min(WKT, p.KHp - t0);
  const int a0 = atile * 128;

  const h16_t* Ag = reinterpret_cast<const h16_t*>(p.A);
  const h16_t* Bg = reinterpret_cast<const h16_t*>(p.B);
  const int total_units = p.nseq * p.Q;
  const int nstages = (total_units + WPOS - 1) / WPOS;
  const int st_begin = blockIdx.y * stages_per_split;
  const int st_end = min(nstages, st_begin + stages_per_split);
  if (st_begin >= st_end) return;

  // ---- DMA roles: wave w stages rows [16w, 16w+16) of every tile ----
  // A: 4 instructions of 4 rows x 16 slots;  B: one instruction per tap, 16 rows x 4 slots
  const h16_t* zsrc = reinterpret_cast<const h16_t*>(g_zero_page);
  int arow[4], acol[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    arow[i] = wave * 16 + i * 4 + (lane >> 4);
    const int f = 2 * ((arow[i] & 3) | (((arow[i] >> 3) & 1) << 2));
    acol[i] = a0 + (((lane & 15) ^ f) * 8);
  }
  const int brow = wave * 16 + (lane >> 2);
  const int bcol = ch * 32 + (((lane & 3) ^ (2 * ((brow >> 3) & 1))) * 8);

  auto issue = [&](int st, int buf) {
    unsigned char* base = smem + buf * WSTAGE;
    const int u0 = st * WPOS;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int u = u0 + arow[i];
      const h16_t* src = u < total_units ? Ag + (long)u * p.CA + acol[i] : zsrc;
      glds16(src, base + wave * 4096 + i * 1024);
    }
    const int u = u0 + brow;
    const bool uok = u < total_units;
    const int seq = uok ? u / p.Q : 0;
    const int q = u - seq * p.Q;
    const int r0 = q * p.s + t0 * p.dil + p.off;
    const h16_t* rsrc = Bg + ((long)seq * p.LB + r0) * p.CB + bcol;
#pragma unroll
    for (int t = 0; t < WKT; ++t) {
      const int r = r0 + t * p.dil;
      const bool ok = uok && (unsigned)r < (unsigned)p.LB;
      const h16_t* src = ok ? rsrc + (long)t * p.dil * p.CB : zsrc;
      glds16(src, base + WA_BYTES + t * WB_BYTES + wave * 1024);
    }
  };

  f32x4 acc[4][WKT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < WKT; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // dbias (column sums of A = dy) by the (chunk 0, tap group 0) blocks, read back from the staged tiles
  const bool do_bias = p.dbias != nullptr && ch == 0 && tgi == 0;          // block-uniform
  f32x4 bacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) bacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read addresses (bytes, relative to the stage base) for ks = 0; ks = 1 adds 32 rows
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const int frow = g8 * 8 + (j16 >> 2);                         // + ks*32 (+4 for the high half)
  const int fa = 2 * ((frow & 3) | (((frow >> 3) & 1) << 2));   // unchanged by +4 and +32
  const int fb = 2 * ((frow >> 3) & 1);
  const int half = (j16 & 1) * 8;
  int a_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) a_off[i] = frow * 256 + (((wr * 8 + i * 2 + ((j16 & 3) >> 1)) ^ fa) * 16) + half;
  const int b_off = WA_BYTES + frow * 64 + (((wc * 2 + ((j16 & 3) >> 1)) ^ fb) * 16) + half;

  issue(st_begin, 0);
  for (int st = st_begin; st < st_end; ++st) {
    const int buf = (st - st_begin) & 1;
    __syncthreads();
    if (st + 1 < st_end) issue(st + 1, buf ^ 1);
    const unsigned sbase = lds0 + buf * WSTAGE;
    unsigned aa[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) aa[i] = sbase + a_off[i];
    const unsigned ba = sbase + b_off;
    h16x8 a[4], b[WKT];
    tr_load_step<0>(aa, ba, a, b);
#pragma unroll
    for (int t = 0; t < WKT; ++t)
      if (t < ntap)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][t] = EVT_MFMA_16x16x32(a[i], b[t], acc[i][t], 0, 0, 0);
    if (do_bias) wg_bias_mma<4>(bacc, a, wc);
    tr_load_step<1>(aa, ba, a, b);
#pragma unroll
    for (int t = 0; t < WKT; ++t)
      if (t < ntap)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][t] = EVT_MFMA_16x16x32(a[i], b[t], acc[i][t], 0, 0, 0);
    if (do_bias) wg_bias_mma<4>(bacc, a, wc);
  }
  if (do_bias) wg_finish_bias_mma<4>(p, bacc, a0, wr, wc, g8, j16, blockIdx.y);
  // lane holds A channels g8*4..+3 (rows) x B channel j16 (column) of each tile
  wg_finish<4, WKT>(p, smem, acc, ntap, a0, ch, t0, wr, wc, g8, j16, blockIdx.y);
}

// -----------------------------------------------------------------------------------------------------------------
// wgrad_gemm: the weight gradient of a dense layer (k = 1), dW[N][K] += dy[M][N]^T . x[M][K], as a 128 x 128 GEMM tile.
// Same structure as wgrad_deep -- both operands position-major, LDS-DMA staging, ds_read_b64_tr_b16 fragments, K stage of
// 64 positions, split over blockIdx.y with fp32 atomics -- but the B tile is FOUR 32-channel chunks of the SAME rows
// instead of five taps of one chunk: a 5-tap x 32-channel tile degenerates to 128 x 32 for k = 1 (the s1 Linear layers
// ran at 117 TFLOP/s on wgrad_ring<4, 1, 4> for that reason).  A wave owns 64 dy-channels x (16 x-channels of each of
// the 4 chunks): 4 x 4 MFMA tiles per K = 32 step, 16 transpose reads behind one wait.
// Reference call sites: the backward of F.linear at transformer.py:207-224,330-334, patched_mha_with_cache.py:242,460.
// -----------------------------------------------------------------------------------------------------------------
constexpr int GKT = 4;                                   // 32-channel chunks of x per block (128 channels)
constexpr int GSTAGE = WA_BYTES + GKT * WB_BYTES;        // 32 KiB

template <int KS>
__device__ __forceinline__ void tr_load_step_g(const unsigned (&aa)[4], unsigned ba, h16x8 (&a)[4], h16x8 (&b)[GKT]) {
  uint2 al[4], ah[4], bl[GKT], bh[GKT];
  if constexpr (KS == 0) {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %16 offset:0\n\t"
        "ds_read_b64_tr_b16 %4, %16 offset:1024\n\t"
        "ds_read_b64_tr_b16 %1, %17 offset:0\n\t"
        "ds_read_b64_tr_b16 %5, %17 offset:1024\n\t"
        "ds_read_b64_tr_b16 %2, %18 offset:0\n\t"
        "ds_read_b64_tr_b16 %6, %18 offset:1024\n\t"
        "ds_read_b64_tr_b16 %3, %19 offset:0\n\t"
        "ds_read_b64_tr_b16 %7, %19 offset:1024\n\t"
        "ds_read_b64_tr_b16 %8, %20 offset:0\n\t"
        "ds_read_b64_tr_b16 %12, %20 offset:256\n\t"
        "ds_read_b64_tr_b16 %9, %20 offset:4096\n\t"
        "ds_read_b64_tr_b16 %13, %20 offset:4352\n\t"
        "ds_read_b64_tr_b16 %10, %20 offset:8192\n\t"
        "ds_read_b64_tr_b16 %14, %20 offset:8448\n\t"
        "ds_read_b64_tr_b16 %11, %20 offset:12288\n\t"
        "ds_read_b64_tr_b16 %15, %20 offset:12544\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : "=&v"(al[0]), "=&v"(al[1]), "=&v"(al[2]), "=&v"(al[3]), "=&v"(ah[0]), "=&v"(ah[1]), "=&v"(ah[2]), "=&v"(ah[3]),
          "=&v"(bl[0]), "=&v"(bl[1]), "=&v"(bl[2]), "=&v"(bl[3]), "=&v"(bh[0]), "=&v"(bh[1]), "=&v"(bh[2]), "=&v"(bh[3])
        : "v"(aa[0]), "v"(aa[1]), "v"(aa[2]), "v"(aa[3]), "v"(ba)
        : "memory");
  } else {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %16 offset:8192\n\t"
        "ds_read_b64_tr_b16 %4, %16 offset:9216\n\t"
        "ds_read_b64_tr_b16 %1, %17 offset:8192\n\t"
        "ds_read_b64_tr_b16 %5, %17 offset:9216\n\t"
        "ds_read_b64_tr_b16 %2, %18 offset:8192\n\t"
        "ds_read_b64_tr_b16 %6, %18 offset:9216\n\t"
        "ds_read_b64_tr_b16 %3, %19 offset:8192\n\t"
        "ds_read_b64_tr_b16 %7, %19 offset:9216\n\t"
        "ds_read_b64_tr_b16 %8, %20 offset:2048\n\t"
        "ds_read_b64_tr_b16 %12, %20 offset:2304\n\t"
        "ds_read_b64_tr_b16 %9, %20 offset:6144\n\t"
        "ds_read_b64_tr_b16 %13, %20 offset:6400\n\t"
        "ds_read_b64_tr_b16 %10, %20 offset:10240\n\t"
        "ds_read_b64_tr_b16 %14, %20 offset:10496\n\t"
        "ds_read_b64_tr_b16 %11, %20 offset:14336\n\t"
        "ds_read_b64_tr_b16 %15, %20 offset:14592\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : "=&v"(al[0]), "=&v"(al[1]), "=&v"(al[2]), "=&v"(al[3]), "=&v"(ah[0]), "=&v"(ah[1]), "=&v"(ah[2]), "=&v"(ah[3]),
          "=&v"(bl[0]), "=&v"(bl[1]), "=&v"(bl[2]), "=&v"(bl[3]), "=&v"(bh[0]), "=&v"(bh[1]), "=&v"(bh[2]), "=&v"(bh[3])
        : "v"(aa[0]), "v"(aa[1]), "v"(aa[2]), "v"(aa[3]), "v"(ba)
        : "memory");
  }
  union { uint4 u; h16x8 v; } r;
#pragma unroll
  for (int i = 0; i < 4; ++i) { r.u = make_uint4(al[i].x, al[i].y, ah[i].x, ah[i].y); a[i] = r.v; }
#pragma unroll
  for (int t = 0; t < GKT; ++t) { r.u = make_uint4(bl[t].x, bl[t].y, bh[t].x, bh[t].y); b[t] = r.v; }
}

// NS: stages of the LDS ring (32 KiB each); the product launches NS = 2 (two blocks per CU), see launch_wgrad_gemm_grid.
template <int NS>
__global__ __launch_bounds__(256, NS > 2 ? 1 : 2) void wgrad_gemm(WgP p, int stages_per_split) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j16 = lane & 15, g8 = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;

  // Block -> (tile, split).  Consecutive workgroup ids go round-robin over the 8 XCDs, so with the plain (x = tile,
  // y = split) grid the tiles of ONE split -- the blocks that read the same token rows -- sit on all eight L2s and every
  // L2 pulls (nearly) every operand row across the fabric.  xcd_order: the linear id is decoded so that a split lives on
  // one XCD (split = xcd + 8 * ...), its tiles dispatched back to back: an operand row enters one L2, once.
  int tile = blockIdx.x, split = blockIdx.y;
  if (p.xcd_order) {
    const int lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int xcd = lin & 7, idx = lin >> 3;
    tile = idx % (int)gridDim.x;
    split = xcd + 8 * (idx / (int)gridDim.x);
  }
  const int ngrp = p.nchunk / GKT;                       // 128-channel groups of x
  const int cg = tile % ngrp;
  const int atile = tile / ngrp;
  const int a0 = atile * 128;

  const h16_t* Ag = reinterpret_cast<const h16_t*>(p.A);
  const h16_t* Bg = reinterpret_cast<const h16_t*>(p.B);
  const int total_units = p.nseq * p.Q;
  const int nstages = (total_units + WPOS - 1) / WPOS;
  const int st_begin = split * stages_per_split;
  const int st_end = min(nstages, st_begin + stages_per_split);
  if (st_begin >= st_end) return;

  // DMA roles as in wgrad_deep: wave w stages rows [16w, 16w+16) of every tile
  const h16_t* zsrc = reinterpret_cast<const h16_t*>(g_zero_page);
  int arow[4], acol[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    arow[i] = wave * 16 + i * 4 + (lane >> 4);
    const int f = 2 * ((arow[i] & 3) | (((arow[i] >> 3) & 1) << 2));
    acol[i] = a0 + (((lane & 15) ^ f) * 8);
  }
  const int brow = wave * 16 + (lane >> 2);
  const int bcol = cg * 128 + (((lane & 3) ^ (2 * ((brow >> 3) & 1))) * 8);      // + 32 * t per chunk

  auto issue = [&](int st, int buf) {
    unsigned char* base = smem + buf * GSTAGE;
    const int u0 = st * WPOS;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int u = u0 + arow[i];
      const h16_t* src = u < total_units ? Ag + (long)u * p.CA + acol[i] : zsrc;
      glds16(src, base + wave * 4096 + i * 1024);
    }
    const int u = u0 + brow;
    const bool uok = u < total_units;
    const h16_t* rsrc = Bg + (long)(uok ? u : 0) * p.CB + bcol;
#pragma unroll
    for (int t = 0; t < GKT; ++t) glds16(uok ? rsrc + t * 32 : zsrc, base + WA_BYTES + t * WB_BYTES + wave * 1024);
  };

  f32x4 acc[4][GKT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < GKT; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // dbias (column sums of dy) on the matrix pipe (wgrad_epi.h), spread over the blocks that read the same dy tile: block
  // cg takes the fragments i = cg (mod bgrp) in its wc = 0 waves -- one extra MFMA per 16 where ngrp >= 4, no block of the
  // grid slower than the others (all four fragments in the cg = 0 blocks: 657 TFLOP/s at [32768, 2048, 512], 763 without)
  const int bgrp = ngrp < 4 ? ngrp : 4;
  const bool do_bias = p.dbias != nullptr && cg < bgrp && wc == 0;         // wave-uniform
  bool mine[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) mine[i] = do_bias && (i % bgrp) == cg;
  h16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (evt_hn)1.0f;
  f32x4 bacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) bacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const int frow = g8 * 8 + (j16 >> 2);
  const int fa = 2 * ((frow & 3) | (((frow >> 3) & 1) << 2));
  const int fb = 2 * ((frow >> 3) & 1);
  const int half = (j16 & 1) * 8;
  int a_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) a_off[i] = frow * 256 + (((wr * 8 + i * 2 + ((j16 & 3) >> 1)) ^ fa) * 16) + half;
  const int b_off = WA_BYTES + frow * 64 + (((wc * 2 + ((j16 & 3) >> 1)) ^ fb) * 16) + half;

  const int nst = st_end - st_begin;
  constexpr int G = 4 + GKT;                               // DMA instructions per wave per stage
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nst) issue(st_begin + s, s);
  for (int st = st_begin; st < st_end; ++st) {
    const int s = st - st_begin;
    const int buf = s % NS;
    // groups still allowed in flight once stage s has landed: stages s+1 .. s+NS-2 that exist
    wait_groups<G, NS - 2>(min(NS - 2, nst - 1 - s));
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (s + NS - 1 < nst) issue(st + NS - 1, (s + NS - 1) % NS);
    const unsigned sbase = lds0 + buf * GSTAGE;
    unsigned aa[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) aa[i] = sbase + a_off[i];
    const unsigned ba = sbase + b_off;
    h16x8 a[4], b[GKT];
    tr_load_step_g<0>(aa, ba, a, b);
#pragma unroll
    for (int t = 0; t < GKT; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][t] = EVT_MFMA_16x16x32(a[i], b[t], acc[i][t], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (mine[i]) bacc[i] = EVT_MFMA_16x16x32(a[i], ones, bacc[i], 0, 0, 0);
    tr_load_step_g<1>(aa, ba, a, b);
#pragma unroll
    for (int t = 0; t < GKT; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][t] = EVT_MFMA_16x16x32(a[i], b[t], acc[i][t], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (mine[i]) bacc[i] = EVT_MFMA_16x16x32(a[i], ones, bacc[i], 0, 0, 0);
    asm volatile("" ::: "memory");
  }
  if (do_bias && j16 == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (mine[i])
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(p.dbias + a0 + wr * 64 + i * 16 + g8 * 4 + r, bacc[i][r]);
  }

  if (p.parts < 0) {      // measurement variant (EVT_WGRAD_GEMM_NOEPI=1): the main loop without its atomics
    if (acc[0][0][0] == 12345.678f) p.dw[0] = 1.f;
    return;
  }
  // lane holds dy-channels g8*4..+3 (rows) x x-channel j16 (column) of each tile; image [CA][nchunk][1][32]
#pragma unroll
  for (int t = 0; t < GKT; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int a = a0 + wr * 64 + i * 16 + g8 * 4 + r;
        const long off = ((long)a * p.nchunk + cg * GKT + t) * 32 + wc * 16 + j16;
        atomicAdd(p.dw + off, acc[i][t][r]);
      }
}

// -----------------------------------------------------------------------------------------------------------------
// wgrad_ring<MA, KT, NS>: wgrad_deep's GEMM with a block tile of 32*MA A-channels x (KT taps x 32 B-channels) and an
// NS-deep ring of 64-position K stages with counted waits and one raw barrier per stage (see conv_ring): the weight
// gradients of the WN / FFN / mid-width vocoder layers are 3200..20000-position reductions that were bound by one
// HBM round trip per stage.  TrStep<MA, KT>: all transpose reads of one K = 32 step behind one wait (generated per
// shape because the row / tap / k-step displacements are instruction offsets).
// A-tile swizzle (16-byte slots, applied on the DMA source and undone on the read):
//   MA = 4 (256-byte rows): slot ^= 2*((row&3) | ((row>>3)&1)<<2)      MA = 2 (128-byte rows): slot ^= 2*(((row>>1)&1) | ((row>>3)&1)<<1)
// -----------------------------------------------------------------------------------------------------------------
template <int MA, int KT> struct TrStep;
template <> struct TrStep<2, 1> {
  template <int KS>
  static __device__ __forceinline__ void load(const unsigned (&aa)[2], unsigned ba, h16x8 (&a)[2], h16x8 (&b)[1]) {
    uint2 al[2], ah[2], bl[1], bh[1];
    if constexpr (KS == 0) {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %6 offset:0\n\t"
        "ds_read_b64_tr_b16 %2, %6 offset:512\n\t"
        "ds_read_b64_tr_b16 %1, %7 offset:0\n\t"
        "ds_read_b64_tr_b16 %3, %7 offset:512\n\t"
        "ds_read_b64_tr_b16 %4, %8 offset:0\n\t"
        "ds_read_b64_tr_b16 %5, %8 offset:256\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : "=&v"(al[0]), "=&v"(al[1]), "=&v"(ah[0]), "=&v"(ah[1]), "=&v"(bl[0]), "=&v"(bh[0])
        : "v"(aa[0]), "v"(aa[1]), "v"(ba)
        : "memory");
    } else {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %6 offset:4096\n\t"
        "ds_read_b64_tr_b16 %2, %6 offset:4608\n\t"
        "ds_read_b64_tr_b16 %1, %7 offset:4096\n\t"
        "ds_read_b64_tr_b16 %3, %7 offset:4608\n\t"
        "ds_read_b64_tr_b16 %4, %8 offset:2048\n\t"
        "ds_read_b64_tr_b16 %5, %8 offset:2304\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : "=&v"(al[0]), "=&v"(al[1]), "=&v"(ah[0]), "=&v"(ah[1]), "=&v"(bl[0]), "=&v"(bh[0])
        : "v"(aa[0]), "v"(aa[1]), "v"(ba)
        : "memory");
    }
    union { uint4 u; h16x8 v; } r;
#pragma unroll
    for (int i = 0; i < 2; ++i) { r.u = make_uint4(al[i].x, al[i].y, ah[i].x, ah[i].y); a[i] = r.v; }
#pragma unroll
    for (int t = 0; t < 1; ++t) { r.u = make_uint4(bl[t].x, bl[t].y, bh[t].x, bh[t].y); b[t] = r.v; }
  }
};
template <> struct TrStep<2, 3> {
  template <int KS>
  static __device__ __forceinline__ void load(const unsigned (&aa)[2], unsigned ba, h16x8 (&a)[2], h16x8 (&b)[3]) {
    uint2 al[2], ah[2], bl[3], bh[3];
    if constexpr (KS == 0) {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %10 offset:0\n\t"
        "ds_read_b64_tr_b16 %2, %10 offset:512\n\t"
        "ds_read_b64_tr_b16 %1, %11 offset:0\n\t"
        "ds_read_b64_tr_b16 %3, %11 offset:512\n\t"
        "ds_read_b64_tr_b16 %4, %12 offset:0\n\t"
        "ds_read_b64_tr_b16 %7, %12 offset:256\n\t"
        "ds_read_b64_tr_b16 %5, %12 offset:4096\n\t"
        "ds_read_b64_tr_b16 %8, %12 offset:4352\n\t"
        "ds_read_b64_tr_b16 %6, %12 offset:8192\n\t"
        "ds_read_b64_tr_b16 %9, %12 offset:8448\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : "=&v"(al[0]), "=&v"(al[1]), "=&v"(ah[0]), "=&v"(ah[1]), "=&v"(bl[0]), "=&v"(bl[1]), "=&v"(bl[2]), "=&v"(bh[0]), "=&v"(bh[1]), "=&v"(bh[2])
        : "v"(aa[0]), "v"(aa[1]), "v"(ba)
        : "memory");
    } else {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %10 offset:4096\n\t"
        "ds_read_b64_tr_b16 %2, %10 offset:4608\n\t"
        "ds_read_b64_tr_b16 %1, %11 offset:4096\n\t"
        "ds_read_b64_tr_b16 %3, %11 offset:4608\n\t"
        "ds_read_b64_tr_b16 %4, %12 offset:2048\n\t"
        "ds_read_b64_tr_b16 %7, %12 offset:2304\n\t"
        "ds_read_b64_tr_b16 %5, %12 offset:6144\n\t"
        "ds_read_b64_tr_b16 %8, %12 offset:6400\n\t"
        "ds_read_b64_tr_b16 %6, %12 offset:10240\n\t"
        "ds_read_b64_tr_b16 %9, %12 offset:10496\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : "=&v"(al[0]), "=&v"(al[1]), "=&v"(ah[0]), "=&v"(ah[1]), "=&v"(bl[0]), "=&v"(bl[1]), "=&v"(bl[2]), "=&v"(bh[0]), "=&v"(bh[1]), "=&v"(bh[2])
        : "v"(aa[0]), "v"(aa[1]), "v"(ba)
        : "memory");
    }
    union { uint4 u; h16x8 v; } r;
#pragma unroll
    for (int i = 0; i < 2; ++i) { r.u = make_uint4(al[i].x, al[i].y, ah[i].x, ah[i].y); a[i] = r.v; }
#pragma unroll
    for (int t = 0; t < 3; ++t) { r.u = make_uint4(bl[t].x, bl[t].y, bh[t].x, bh[t].y); b[t] = r.v; }
  }
};
template <> struct TrStep<2, 5> {
  template <int KS>
  static __device__ __forceinline__ void load(const unsigned (&aa)[2], unsigned ba, h16x8 (&a)[2], h16x8 (&b)[5]) {
    uint2 al[2], ah[2], bl[5], bh[5];
    if constexpr (KS == 0) {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %14 offset:0\n\t"
        "ds_read_b64_tr_b16 %2, %14 offset:512\n\t"
        "ds_read_b64_tr_b16 %1, %15 offset:0\n\t"
        "ds_read_b64_tr_b16 %3, %15 offset:512\n\t"
        "ds_read_b64_tr_b16 %4, %16 offset:0\n\t"
        "ds_read_b64_tr_b16 %9, %16 offset:256\n\t"
        "ds_read_b64_tr_b16 %5, %16 offset:4096\n\t"
        "ds_read_b64_tr_b16 %10, %16 offset:4352\n\t"
        "ds_read_b64_tr_b16 %6, %16 offset:8192\n\t"
        "ds_read_b64_tr_b16 %11, %16 offset:8448\n\t"
        "ds_read_b64_tr_b16 %7, %16 offset:12288\n\t"
        "ds_read_b64_tr_b16 %12, %16 offset:12544\n\t"
        "ds_read_b64_tr_b16 %8, %16 offset:16384\n\t"
        "ds_read_b64_tr_b16 %13, %16 offset:16640\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : "=&v"(al[0]), "=&v"(al[1]), "=&v"(ah[0]), "=&v"(ah[1]), "=&v"(bl[0]), "=&v"(bl[1]), "=&v"(bl[2]), "=&v"(bl[3]), "=&v"(bl[4]), "=&v"(bh[0]), "=&v"(bh[1]), "=&v"(bh[2]), "=&v"(bh[3]), "=&v"(bh[4])
        : "v"(aa[0]), "v"(aa[1]), "v"(ba)
        : "memory");
    } else {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %14 offset:4096\n\t"
        "ds_read_b64_tr_b16 %2, %14 offset:4608\n\t"
        "ds_read_b64_tr_b16 %1, %15 offset:4096\n\t"
        "ds_read_b64_tr_b16 %3, %15 offset:4608\n\t"
        "ds_read_b64_tr_b16 %4, %16 offset:2048\n\t"
        "ds_read_b64_tr_b16 %9, %16 offset:2304\n\t"
        "ds_read_b64_tr_b16 %5, %16 offset:6144\n\t"
        "ds_read_b64_tr_b16 %10, %16 offset:6400\n\t"
        "ds_read_b64_tr_b16 %6, %16 offset:10240\n\t"
        "ds_read_b64_tr_b16 %11, %16 offset:10496\n\t"
        "ds_read_b64_tr_b16 %7, %16 offset:14336\n\t"
        "ds_read_b64_tr_b16 %12, %16 offset:14592\n\t"
        "ds_read_b64_tr_b16 %8, %16 offset:18432\n\t"
        "ds_read_b64_tr_b16 %13, %16 offset:18688\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : "=&v"(al[0]), "=&v"(al[1]), "=&v"(ah[0]), "=&v"(ah[1]), "=&v"(bl[0]), "=&v"(bl[1]), "=&v"(bl[2]), "=&v"(bl[3]), "=&v"(bl[4]), "=&v"(bh[0]), "=&v"(bh[1]), "=&v"(bh[2]), "=&v"(bh[3]), "=&v"(bh[4])
        : "v"(aa[0]), "v"(aa[1]), "v"(ba)
        : "memory");
    }
    union { uint4 u; h16x8 v; } r;
#pragma unroll
    for (int i = 0; i < 2; ++i) { r.u = make_uint4(al[i].x, al[i].y, ah[i].x, ah[i].y); a[i] = r.v; }
#pragma unroll
    for (int t = 0; t < 5; ++t) { r.u = make_uint4(bl[t].x, bl[t].y, bh[t].x, bh[t].y); b[t] = r.v; }
  }
};
template <> struct TrStep<4, 1> {
  template <int KS>
  static __device__ __forceinline__ void load(const unsigned (&aa)[4], unsigned ba, h16x8 (&a)[4], h16x8 (&b)[1]) {
    uint2 al[4], ah[4], bl[1], bh[1];
    if constexpr (KS == 0) {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %10 offset:0\n\t"
        "ds_read_b64_tr_b16 %4, %10 offset:1024\n\t"
        "ds_read_b64_tr_b16 %1, %11 offset:0\n\t"
        "ds_read_b64_tr_b16 %5, %11 offset:1024\n\t"
        "ds_read_b64_tr_b16 %2, %12 offset:0\n\t"
        "ds_read_b64_tr_b16 %6, %12 offset:1024\n\t"
        "ds_read_b64_tr_b16 %3, %13 offset:0\n\t"
        "ds_read_b64_tr_b16 %7, %13 offset:1024\n\t"
        "ds_read_b64_tr_b16 %8, %14 offset:0\n\t"
        "ds_read_b64_tr_b16 %9, %14 offset:256\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : "=&v"(al[0]), "=&v"(al[1]), "=&v"(al[2]), "=&v"(al[3]), "=&v"(ah[0]), "=&v"(ah[1]), "=&v"(ah[2]), "=&v"(ah[3]), "=&v"(bl[0]), "=&v"(bh[0])
        : "v"(aa[0]), "v"(aa[1]), "v"(aa[2]), "v"(aa[3]), "v"(ba)
        : "memory");
    } else {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %10 offset:8192\n\t"
        "ds_read_b64_tr_b16 %4, %10 offset:9216\n\t"
        "ds_read_b64_tr_b16 %1, %11 offset:8192\n\t"
        "ds_read_b64_tr_b16 %5, %11 offset:9216\n\t"
        "ds_read_b64_tr_b16 %2, %12 offset:8192\n\t"
        "ds_read_b64_tr_b16 %6, %12 offset:9216\n\t"
        "ds_read_b64_tr_b16 %3, %13 offset:8192\n\t"
        "ds_read_b64_tr_b16 %7, %13 offset:9216\n\t"
        "ds_read_b64_tr_b16 %8, %14 offset:2048\n\t"
        "ds_read_b64_tr_b16 %9, %14 offset:2304\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : "=&v"(al[0]), "=&v"(al[1]), "=&v"(al[2]), "=&v"(al[3]), "=&v"(ah[0]), "=&v"(ah[1]), "=&v"(ah[2]), "=&v"(ah[3]), "=&v"(bl[0]), "=&v"(bh[0])
        : "v"(aa[0]), "v"(aa[1]), "v"(aa[2]), "v"(aa[3]), "v"(ba)
        : "memory");
    }
    union { uint4 u; h16x8 v; } r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r.u = make_uint4(al[i].x, al[i].y, ah[i].x, ah[i].y); a[i] = r.v; }
#pragma unroll
    for (int t = 0; t < 1; ++t) { r.u = make_uint4(bl[t].x, bl[t].y, bh[t].x, bh[t].y); b[t] = r.v; }
  }
};
template <> struct TrStep<4, 3> {
  template <int KS>
  static __device__ __forceinline__ void load(const unsigned (&aa)[4], unsigned ba, h16x8 (&a)[4], h16x8 (&b)[3]) {
    uint2 al[4], ah[4], bl[3], bh[3];
    if constexpr (KS == 0) {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %14 offset:0\n\t"
        "ds_read_b64_tr_b16 %4, %14 offset:1024\n\t"
        "ds_read_b64_tr_b16 %1, %15 offset:0\n\t"
        "ds_read_b64_tr_b16 %5, %15 offset:1024\n\t"
        "ds_read_b64_tr_b16 %2, %16 offset:0\n\t"
        "ds_read_b64_tr_b16 %6, %16 offset:1024\n\t"
        "ds_read_b64_tr_b16 %3, %17 offset:0\n\t"
        "ds_read_b64_tr_b16 %7, %17 offset:1024\n\t"
        "ds_read_b64_tr_b16 %8, %18 offset:0\n\t"
        "ds_read_b64_tr_b16 %11, %18 offset:256\n\t"
        "ds_read_b64_tr_b16 %9, %18 offset:4096\n\t"
        "ds_read_b64_tr_b16 %12, %18 offset:4352\n\t"
        "ds_read_b64_tr_b16 %10, %18 offset:8192\n\t"
        "ds_read_b64_tr_b16 %13, %18 offset:8448\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : "=&v"(al[0]), "=&v"(al[1]), "=&v"(al[2]), "=&v"(al[3]), "=&v"(ah[0]), "=&v"(ah[1]), "=&v"(ah[2]), "=&v"(ah[3]), "=&v"(bl[0]), "=&v"(bl[1]), "=&v"(bl[2]), "=&v"(bh[0]), "=&v"(bh[1]), "=&v"(bh[2])
        : "v"(aa[0]), "v"(aa[1]), "v"(aa[2]), "v"(aa[3]), "v"(ba)
        : "memory");
    } else {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %14 offset:8192\n\t"
        "ds_read_b64_tr_b16 %4, %14 offset:9216\n\t"
        "ds_read_b64_tr_b16 %1, %15 offset:8192\n\t"
        "ds_read_b64_tr_b16 %5, %15 offset:9216\n\t"
        "ds_read_b64_tr_b16 %2, %16 offset:8192\n\t"
        "ds_read_b64_tr_b16 %6, %16 offset:9216\n\t"
        "ds_read_b64_tr_b16 %3, %17 offset:8192\n\t"
        "ds_read_b64_tr_b16 %7, %17 offset:9216\n\t"
        "ds_read_b64_tr_b16 %8, %18 offset:2048\n\t"
        "ds_read_b64_tr_b16 %11, %18 offset:2304\n\t"
        "ds_read_b64_tr_b16 %9, %18 offset:6144\n\t"
        "ds_read_b64_tr_b16 %12, %18 offset:6400\n\t"
        "ds_read_b64_tr_b16 %10, %18 offset:10240\n\t"
        "ds_read_b64_tr_b16 %13, %18 offset:10496\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : "=&v"(al[0]), "=&v"(al[1]), "=&v"(al[2]), "=&v"(al[3]), "=&v"(ah[0]), "=&v"(ah[1]), "=&v"(ah[2]), "=&v"(ah[3]), "=&v"(bl[0]), "=&v"(bl[1]), "=&v"(bl[2]), "=&v"(bh[0]), "=&v"(bh[1]), "=&v"(bh[2])
        : "v"(aa[0]), "v"(aa[1]), "v"(aa[2]), "v"(aa[3]), "v"(ba)
        : "memory");
    }
    union { uint4 u; h16x8 v; } r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r.u = make_uint4(al[i].x, al[i].y, ah[i].x, ah[i].y); a[i] = r.v; }
#pragma unroll
    for (int t = 0; t < 3; ++t) { r.u = make_uint4(bl[t].x, bl[t].y, bh[t].x, bh[t].y); b[t] = r.v; }
  }
};
template <> struct TrStep<4, 5> {
  template <int KS>
  static __device__ __forceinline__ void load(const unsigned (&aa)[4], unsigned ba, h16x8 (&a)[4], h16x8 (&b)[5]) {
    uint2 al[4], ah[4], bl[5], bh[5];
    if constexpr (KS == 0) {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %18 offset:0\n\t"
        "ds_read_b64_tr_b16 %4, %18 offset:1024\n\t"
        "ds_read_b64_tr_b16 %1, %19 offset:0\n\t"
        "ds_read_b64_tr_b16 %5, %19 offset:1024\n\t"
        "ds_read_b64_tr_b16 %2, %20 offset:0\n\t"
        "ds_read_b64_tr_b16 %6, %20 offset:1024\n\t"
        "ds_read_b64_tr_b16 %3, %21 offset:0\n\t"
        "ds_read_b64_tr_b16 %7, %21 offset:1024\n\t"
        "ds_read_b64_tr_b16 %8, %22 offset:0\n\t"
        "ds_read_b64_tr_b16 %13, %22 offset:256\n\t"
        "ds_read_b64_tr_b16 %9, %22 offset:4096\n\t"
        "ds_read_b64_tr_b16 %14, %22 offset:4352\n\t"
        "ds_read_b64_tr_b16 %10, %22 offset:8192\n\t"
        "ds_read_b64_tr_b16 %15, %22 offset:8448\n\t"
        "ds_read_b64_tr_b16 %11, %22 offset:12288\n\t"
        "ds_read_b64_tr_b16 %16, %22 offset:12544\n\t"
        "ds_read_b64_tr_b16 %12, %22 offset:16384\n\t"
        "ds_read_b64_tr_b16 %17, %22 offset:16640\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : "=&v"(al[0]), "=&v"(al[1]), "=&v"(al[2]), "=&v"(al[3]), "=&v"(ah[0]), "=&v"(ah[1]), "=&v"(ah[2]), "=&v"(ah[3]), "=&v"(bl[0]), "=&v"(bl[1]), "=&v"(bl[2]), "=&v"(bl[3]), "=&v"(bl[4]), "=&v"(bh[0]), "=&v"(bh[1]), "=&v"(bh[2]), "=&v"(bh[3]), "=&v"(bh[4])
        : "v"(aa[0]), "v"(aa[1]), "v"(aa[2]), "v"(aa[3]), "v"(ba)
        : "memory");
    } else {
    asm volatile(
        "ds_read_b64_tr_b16 %0, %18 offset:8192\n\t"
        "ds_read_b64_tr_b16 %4, %18 offset:9216\n\t"
        "ds_read_b64_tr_b16 %1, %19 offset:8192\n\t"
        "ds_read_b64_tr_b16 %5, %19 offset:9216\n\t"
        "ds_read_b64_tr_b16 %2, %20 offset:8192\n\t"
        "ds_read_b64_tr_b16 %6, %20 offset:9216\n\t"
        "ds_read_b64_tr_b16 %3, %21 offset:8192\n\t"
        "ds_read_b64_tr_b16 %7, %21 offset:9216\n\t"
        "ds_read_b64_tr_b16 %8, %22 offset:2048\n\t"
        "ds_read_b64_tr_b16 %13, %22 offset:2304\n\t"
        "ds_read_b64_tr_b16 %9, %22 offset:6144\n\t"
        "ds_read_b64_tr_b16 %14, %22 offset:6400\n\t"
        "ds_read_b64_tr_b16 %10, %22 offset:10240\n\t"
        "ds_read_b64_tr_b16 %15, %22 offset:10496\n\t"
        "ds_read_b64_tr_b16 %11, %22 offset:14336\n\t"
        "ds_read_b64_tr_b16 %16, %22 offset:14592\n\t"
        "ds_read_b64_tr_b16 %12, %22 offset:18432\n\t"
        "ds_read_b64_tr_b16 %17, %22 offset:18688\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        : "=&v"(al[0]), "=&v"(al[1]), "=&v"(al[2]), "=&v"(al[3]), "=&v"(ah[0]), "=&v"(ah[1]), "=&v"(ah[2]), "=&v"(ah[3]), "=&v"(bl[0]), "=&v"(bl[1]), "=&v"(bl[2]), "=&v"(bl[3]), "=&v"(bl[4]), "=&v"(bh[0]), "=&v"(bh[1]), "=&v"(bh[2]), "=&v"(bh[3]), "=&v"(bh[4])
        : "v"(aa[0]), "v"(aa[1]), "v"(aa[2]), "v"(aa[3]), "v"(ba)
        : "memory");
    }
    union { uint4 u; h16x8 v; } r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r.u = make_uint4(al[i].x, al[i].y, ah[i].x, ah[i].y); a[i] = r.v; }
#pragma unroll
    for (int t = 0; t < 5; ++t) { r.u = make_uint4(bl[t].x, bl[t].y, bh[t].x, bh[t].y); b[t] = r.v; }
  }
};

template <int MA> __device__ __forceinline__ int a_swz(int row) {
  return MA == 4 ? 2 * ((row & 3) | (((row >> 3) & 1) << 2)) : 2 * (((row >> 1) & 1) | (((row >> 3) & 1) << 1));
}

template <int MA, int KT, int NS>
__global__ __launch_bounds__(256) void wgrad_ring(WgP p, int stages_per_split) {
  constexpr int AROW = 64 * MA;                     // bytes per A-tile row (32*MA channels)
  constexpr int ABYTES = WPOS * AROW;
  constexpr int STAGE = ABYTES + KT * WB_BYTES;
  constexpr int G = MA + KT;                        // DMA instructions per wave per stage
  constexpr int RPI = 16 / MA;                      // A rows per DMA instruction
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j16 = lane & 15, g8 = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;

  // xcd_order (split count a multiple of 8): a split -- one range of positions -- lives on ONE XCD, its tiles dispatched back
  // to back (consecutive workgroup ids go round-robin over the 8 XCDs): the rows of that range enter one L2 instead of
  // all eight (PMC round 3: 34.8 MB per launch against 5 MB of operands for the 192 -> 384 k5 gradient)
  int bx = blockIdx.x, split = blockIdx.y;
  if (p.xcd_order) {
    const int lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int idx = lin >> 3;
    bx = idx % (int)gridDim.x;
    split = (lin & 7) + 8 * (idx / (int)gridDim.x);
  }
  const int ch = bx % p.nchunk; bx /= p.nchunk;
  const int tgi = bx % p.ntapgrp;
  const int atile = bx / p.ntapgrp;
  const int t0 = tgi * KT;
  const int ntap = min(KT, p.KHp - t0);
  const int a0 = atile * 32 * MA;

  const h16_t* Ag = reinterpret_cast<const h16_t*>(p.A);
  const h16_t* Bg = reinterpret_cast<const h16_t*>(p.B);
  const int total_units = p.nseq * p.Q;
  const int nstages = (total_units + WPOS - 1) / WPOS;
  const int st_begin = split * stages_per_split;
  const int nst = min(nstages, st_begin + stages_per_split) - st_begin;
  if (nst <= 0) return;

  const h16_t* zsrc = reinterpret_cast<const h16_t*>(g_zero_page);
  int arow[MA], acol[MA];
#pragma unroll
  for (int i = 0; i < MA; ++i) {
    arow[i] = wave * 16 + i * RPI + lane / (4 * MA);
    acol[i] = a0 + (((lane % (4 * MA)) ^ a_swz<MA>(arow[i])) * 8);
  }
  const int brow = wave * 16 + (lane >> 2);
  const int bcol = ch * 32 + (((lane & 3) ^ (2 * ((brow >> 3) & 1))) * 8);

  auto issue = [&](int s) {
    unsigned char* base = smem + (s % NS) * STAGE;
    const int u0 = (st_begin + s) * WPOS;
#pragma unroll
    for (int i = 0; i < MA; ++i) {
      const int u = u0 + arow[i];
      glds16(u < total_units ? Ag + (long)u * p.CA + acol[i] : zsrc, base + wave * (16 * AROW) + i * 1024);
    }
    const int u = u0 + brow;
    const bool uok = u < total_units;
    const int seq = uok ? u / p.Q : 0;
    const int q = u - seq * p.Q;
    const int r0 = q * p.s + t0 * p.dil + p.off;
    const h16_t* rsrc = Bg + ((long)seq * p.LB + r0) * p.CB + bcol;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      const int r = r0 + t * p.dil;
      const bool ok = uok && (unsigned)r < (unsigned)p.LB;
      glds16(ok ? rsrc + (long)t * p.dil * p.CB : zsrc, base + ABYTES + t * WB_BYTES + wave * 1024);
    }
  };

  f32x4 acc[MA][KT];
#pragma unroll
  for (int i = 0; i < MA; ++i)
#pragma unroll
    for (int t = 0; t < KT; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // dbias (column sums of the A operand = dy) by the (chunk 0, tap group 0) blocks, from the staged tiles
  const bool do_bias = p.dbias != nullptr && ch == 0 && tgi == 0;          // block-uniform
  f32x4 bacc[MA];
#pragma unroll
  for (int i = 0; i < MA; ++i) bacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  const int frow = g8 * 8 + (j16 >> 2);
  const int fa = a_swz<MA>(frow);                   // unchanged by +4 and +32 rows
  const int fb = 2 * ((frow >> 3) & 1);
  const int half = (j16 & 1) * 8;
  int a_off[MA];
#pragma unroll
  for (int i = 0; i < MA; ++i) a_off[i] = frow * AROW + (((wr * 2 * MA + i * 2 + ((j16 & 3) >> 1)) ^ fa) * 16) + half;
  const int b_off = ABYTES + frow * 64 + (((wc * 2 + ((j16 & 3) >> 1)) ^ fb) * 16) + half;

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
    unsigned aa[MA];
#pragma unroll
    for (int i = 0; i < MA; ++i) aa[i] = sbase + a_off[i];
    const unsigned ba = sbase + b_off;
    h16x8 a[MA], b[KT];
    TrStep<MA, KT>::template load<0>(aa, ba, a, b);
#pragma unroll
    for (int t = 0; t < KT; ++t)
      if (t < ntap)
#pragma unroll
        for (int i = 0; i < MA; ++i) acc[i][t] = EVT_MFMA_16x16x32(a[i], b[t], acc[i][t], 0, 0, 0);
    if (do_bias) wg_bias_mma<MA>(bacc, a, wc);
    TrStep<MA, KT>::template load<1>(aa, ba, a, b);
#pragma unroll
    for (int t = 0; t < KT; ++t)
      if (t < ntap)
#pragma unroll
        for (int i = 0; i < MA; ++i) acc[i][t] = EVT_MFMA_16x16x32(a[i], b[t], acc[i][t], 0, 0, 0);
    if (do_bias) wg_bias_mma<MA>(bacc, a, wc);
    asm volatile("" ::: "memory");
  }
  if (do_bias) wg_finish_bias_mma<MA>(p, bacc, a0, wr, wc, g8, j16, split);
  wg_finish<MA, KT>(p, smem, acc, ntap, a0, ch, t0, wr, wc, g8, j16, split);
}

}  // namespace

static int deep_kind(const ConvP& p, int dtype, int out_ch, int k_ch, int nphase);

bool deep_eligible(const ConvP& p, int dtype, int out_ch, int k_ch, int nphase) {
  return deep_kind(p, dtype, out_ch, k_ch, nphase) != 0;
}

// which tile: 2 = 128 x 128 (conv_deep), 1 = 64 x 64 ring (conv_ring<2,2,4>), 0 = not eligible
static int deep_kind(const ConvP& p, int dtype, int out_ch, int k_ch, int nphase) {
  if (dtype != EVT_DT_HALF) return 0;
  if (k_ch % 32 || out_ch % 64) return 0;
  if (p.xact || p.in_slope != 1.f) return 0;              // no load-side fusion on the DMA path
  if (p.nchunk * 32 != k_ch) return 0;                     // prepared image must be the ck = 32 layout
  if ((long)p.nseq * p.Q >= (1L << 31) - BN) return 0;
  const long units = (long)p.nseq * p.Q;
  static const long min128 = getenv("EVT_DEEP_MIN_TILES") ? atol(getenv("EVT_DEEP_MIN_TILES")) : 192;   // tuning knob
  if (out_ch % BM == 0 && ((units + BN - 1) / BN) * (out_ch / BM) * nphase >= min128) return 2;
  static const bool no_ring = getenv("EVT_NO_RING") != nullptr;   // A/B switch for measurements
  if (!no_ring && k_ch % BK == 0 && ((units + 63) / 64) * (out_ch / 64) * nphase >= 32) return 1;
  return 0;
}


int launch_conv_deep(const ConvP& p_in, int out_ch, int k_ch, int nphase, hipStream_t st) {
  ConvP p = p_in;
  const int kind = deep_kind(p, EVT_DT_HALF, out_ch, k_ch, nphase);
  if (kind == 0) return EVT_ENOTSUP;
  if (kind == 1) {
    p.Y = out_ch / 64;
    p.P = (int)(((long)p.nseq * p.Q + 63) / 64);
    p.U = 0;
    // 4 stages: 6 / 8 stages were measured no faster (the 3-36-stage chains cost 0.25-0.5 us per stage; ~10 us of every
    // launch is fixed cost) and leave room for only one block per CU.  Round 5 (profiles/r05_ring_tiles.txt): 64 x 128,
    // 128 x 64 and 128 x 128 tiles and 3 / 6 stages of this kernel on the 3200-position layers -- none faster stand-alone
    // (WN in-layer 12 us forward either way), all slower in the step (23.4 -> 24.0-24.8 ms): fewer, fatter blocks lose more
    // to their own latency chains than the halved block count gives back.  Stand-alone the WN in-layer forward takes 12 us
    // where the step's trace shows 19 us: in the step every layer's 369 KB weight image is read for the first time.
    constexpr int NS = 4;
    static bool attr1 = false;
    const size_t lds1 = (size_t)NS * (64 + 64) * 128;
    if (!attr1) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_ring<2, 2, NS>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1) != hipSuccess)
        return EVT_ELAUNCH;
      attr1 = true;
    }
    evt_set_last_tag("conv_ring<bf16, 64, 64, 64, x4>");
    hipLaunchKernelGGL((conv_ring<2, 2, NS>), dim3(8 * ((p.P + 7) / 8) * p.Y, nphase), dim3(256), lds1, st, p);
    return evt_check_launch();
  }
  p.Y = out_ch / BM;
  p.P = (int)(((long)p.nseq * p.Q + BN - 1) / BN);
  p.U = 0;
  // short reductions (<= 12 stages of 64) and K-side widths that are not multiples of 64: 32-channel stages, twice the
  // resident blocks (measured: 128->512 k5 s3 49 -> 37 us; the 40-80-stage 1024-wide layers are better with 64)
  static const int deep32 = getenv("EVT_DEEP32") ? atoi(getenv("EVT_DEEP32")) : -1;   // A/B switch: 0 never, 1 always
  const bool short_k = (k_ch / 64) * p.KHp <= 12;
  if (deep32 == 1 || k_ch % 64 || (deep32 != 0 && short_k)) {
    static bool attr32 = false;
    const size_t lds32 = 2 * STAGE32 > 128 * 272 ? 2 * STAGE32 : 128 * 272;   // operand stages, then the staged output tile
    if (!attr32) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_deep32), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds32) != hipSuccess)
        return EVT_ELAUNCH;
      attr32 = true;
    }
    evt_set_last_tag("conv_deep32<bf16, 128, 128, 32>");
    hipLaunchKernelGGL(conv_deep32, dim3(8 * ((p.P + 7) / 8) * p.Y, nphase), dim3(256), lds32, st, p);
    return evt_check_launch();
  }
  // Tried in round 4 and dropped: this 128 x 128 tile on the 3- / 4-stage ring of conv_ring (96 / 128 KiB of LDS, ONE block
  // per CU) instead of two double-buffered blocks per CU -- 1024 -> 1024 k5: 104 -> 134-142 us forward, the s2 step
  // 23.8 -> 24.4 / 24.8 ms.  Four waves per CU cannot cover their own fragment-read latency; the second resident block does.
  static bool attr = false;
  const size_t lds = 2 * STAGE_BYTES;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_deep), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return EVT_ELAUNCH;
    attr = true;
  }
  const int gx = 8 * ((p.P + 7) / 8) * p.Y;
  evt_set_last_tag("conv_deep<bf16, 128, 128, 64>");
  hipLaunchKernelGGL(conv_deep, dim3(gx, nphase), dim3(256), lds, st, p);
  return evt_check_launch();
}

void wgrad_pick_split(const WgP& p, long tiles, int nstages, long target, int min_stages, int* nsplit, int* per) {
  long split = (target + tiles - 1) / tiles;
  if (split > nstages / min_stages) split = nstages / min_stages;
  if (p.parts > 0 && split > p.parts) split = p.parts;
  if (split < 1) split = 1;
  *per = (int)((nstages + split - 1) / split);
  *nsplit = (nstages + *per - 1) / *per;          // every split owns at least one stage
}

bool wgrad_deep_eligible(const WgP& p, int dtype) {
  if (dtype != EVT_DT_HALF) return false;
  if (p.CA % 128 || p.CB % 32) return false;
  if (p.Aact || p.Bact || p.a_slope != 1.f || p.b_slope != 1.f) return false;
  if (p.LA != p.Q) return false;                            // A rows are addressed by the flat position
  if (p.KHp < 3) return false;                              // 5-tap tiles: k = 1 / 2 layers go to wgrad_ring<., 1, .>
  if ((long)p.nseq * p.Q >= (1L << 31) - WPOS) return false;
  // worth it only for GEMM-sized problems: enough (A tile, chunk) pairs and enough positions
  const long tiles = (long)(p.CA / 128) * (p.CB / 32) * ((p.KHp + WKT - 1) / WKT);
  const long nstages = ((long)p.nseq * p.Q + WPOS - 1) / WPOS;
  long split = (1024 + tiles - 1) / tiles;
  if (split > nstages / 8) split = nstages / 8;
  return split >= 1 && tiles * split >= 256;                // enough blocks of >= 8 K stages to fill the chip
}

int launch_wgrad_deep(const WgP& p_in, hipStream_t st) {
  WgP p = p_in;
  if (!wgrad_deep_eligible(p, EVT_DT_HALF)) return EVT_ENOTSUP;
  p.nchunk = p.CB / 32;
  p.ntapgrp = (p.KHp + WKT - 1) / WKT;
  const long tiles = (long)(p.CA / 128) * p.nchunk * p.ntapgrp;
  // XCD-aware tile order (EVT_WGRAD_DEEP_XCD=0: plain decode): needs an even number of dy tiles and chunks in fours
  static const bool xcd_on = !(getenv("EVT_WGRAD_DEEP_XCD") && atoi(getenv("EVT_WGRAD_DEEP_XCD")) == 0);
  p.xcd_order = (xcd_on && (p.CA / 128) % 2 == 0 && p.nchunk % 4 == 0) ? 1 : 0;
  const int nstages = (int)(((long)p.nseq * p.Q + WPOS - 1) / WPOS);
  // ~2 blocks per CU in flight, >= 8 K stages per block so the pipeline amortises its fill and the atomics; slab mode:
  // one resident wave of blocks is enough once the tile leaves as plain stores
  int nsplit, per;
  wgrad_pick_split(p, tiles, nstages, p.parts > 0 ? 512 : 1024, 8, &nsplit, &per);
  p.nsplit = nsplit;
  p.now_used = p.prev_used > nsplit ? p.prev_used : nsplit;
  if (p.parts > 0 && p.used_host) *p.used_host = p.now_used;
  static bool attr = false;
  const size_t lds = 2 * WSTAGE;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_deep), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return EVT_ELAUNCH;
    attr = true;
  }
  evt_set_last_tag("wgrad_deep<bf16, 128, 5x32, 64>");
  hipLaunchKernelGGL(wgrad_deep, dim3((unsigned)tiles, p.nsplit), dim3(256), lds, st, p, per);
  return evt_check_launch();
}

bool wgrad_gemm_eligible(const WgP& p, int dtype) {
  static const bool off = getenv("EVT_NO_WGRAD_GEMM") != nullptr;    // A/B switch for measurements
  if (off || dtype != EVT_DT_HALF) return false;
  if (p.KH != 1 || p.KHp != 1 || p.s != 1 || p.off != 0) return false;
  if (p.CA % 128 || p.CB % 128) return false;
  if (p.Aact || p.Bact || p.a_slope != 1.f || p.b_slope != 1.f) return false;
  if (p.LA != p.Q || p.LB != p.Q) return false;             // both operands addressed by the flat position
  const long units = (long)p.nseq * p.Q;
  if (units >= (1L << 31) - WPOS) return false;
  return units >= 2048;                                      // long reductions only: the dense layers of s1
}

static int launch_wgrad_gemm_grid(const WgP& p, long tiles, int per, hipStream_t st);

int launch_wgrad_gemm(const WgP& p_in, hipStream_t st) {
  WgP p = p_in;
  if (!wgrad_gemm_eligible(p, EVT_DT_HALF)) return EVT_ENOTSUP;
  p.nchunk = p.CB / 32;
  p.ntapgrp = 1;
  static const bool noepi = getenv("EVT_WGRAD_GEMM_NOEPI") != nullptr;
  if (noepi) p.parts = -1;
  const long tiles = (long)(p.CA / 128) * (p.CB / 128);
  const int nstages = (int)(((long)p.nseq * p.Q + WPOS - 1) / WPOS);
  // Round 6: HALF a resident wave of blocks (one per CU).  The launch now runs on the s1 engine's side stream next to the
  // backward chain, which fills whatever slots it leaves; what it still pays alone are its fp32 atomics -- memory-side,
  // 1.31 TB/s whatever their scope (profiles/r06_atomics_vs_stores.txt), 15-24 us of a 48-100 us launch at 512 blocks --
  // and half the blocks are half the partial tiles: s1 micro-step 36.43 (512) / 35.91 (384) / 35.83 ms (256) on one box.
  static const long target = getenv("EVT_WGRAD_GEMM_BLOCKS") ? atol(getenv("EVT_WGRAD_GEMM_BLOCKS")) : 256;
  long split = (target + tiles - 1) / tiles;
  if (split > nstages / 8) split = nstages / 8;
  if (split < 1) split = 1;
  const int per = (int)((nstages + split - 1) / split);
  split = (nstages + per - 1) / per;
  // XCD-aware order needs the split count to be a multiple of 8 (EVT_WGRAD_GEMM_XCD=0: the plain grid)
  static const int xcd = getenv("EVT_WGRAD_GEMM_XCD") ? atoi(getenv("EVT_WGRAD_GEMM_XCD")) : 1;
  p.xcd_order = 0;
  if (xcd && nstages >= 64) {
    long s8 = xcd == 2 ? (split / 8 * 8) : (split + 7) / 8 * 8;       // 2: round down (measurement variant)
    if (s8 < 8) s8 = 8;
    const int per8 = (int)((nstages + s8 - 1) / s8);
    if ((long)per8 * (s8 - 1) < nstages) {       // every split non-empty (an empty one only returns, but keep the grid tight)
      split = s8;
      p.xcd_order = 1;
      p.nsplit = (int)split;
      return launch_wgrad_gemm_grid(p, tiles, per8, st);
    }
  }
  p.nsplit = (int)split;
  return launch_wgrad_gemm_grid(p, tiles, per, st);
}

template <int NS>
static int launch_wgrad_gemm_ns(const WgP& p, long tiles, int per, hipStream_t st) {
  static bool attr = false;
  const size_t lds = NS * GSTAGE;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_gemm<NS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return EVT_ELAUNCH;
    attr = true;
  }
  evt_set_last_tag("wgrad_gemm<bf16, 128, 128, 64, x%d>", NS);
  hipLaunchKernelGGL(wgrad_gemm<NS>, dim3((unsigned)tiles, p.nsplit), dim3(256), lds, st, p, per);
  return evt_check_launch();
}

static int launch_wgrad_gemm_grid(const WgP& p, long tiles, int per, hipStream_t st) {
  // NS = 3 / 4 (one block per CU) measured slower than two double-buffered blocks per CU: 367-430 against 600-800 TFLOP/s
  // at [32768, 2048, 512] -- the stage is not waiting for its DMA, a wave is waiting for its own transpose reads, and only
  // a second resident block fills that (PMC: LDS array 12 % busy, no bank conflicts, MFMA 20 %)
  return launch_wgrad_gemm_ns<2>(p, tiles, per, st);
}

// ---- wgrad_ring dispatch ----
template <int MA, int KT, int NS>
static int launch_ring_inst(const WgP& p, int per, hipStream_t st) {
  constexpr size_t lds = (size_t)NS * (WPOS * 64 * MA + KT * WB_BYTES);
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_ring<MA, KT, NS>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return EVT_ELAUNCH;
    attr = true;
  }
  const long tiles = (long)(p.CA / (32 * MA)) * p.nchunk * p.ntapgrp;
  evt_set_last_tag("wgrad_ring<bf16, %d, %dx32, 64, x%d>", 32 * MA, KT, NS);
  hipLaunchKernelGGL((wgrad_ring<MA, KT, NS>), dim3((unsigned)tiles, p.nsplit), dim3(256), lds, st, p, per);
  return evt_check_launch();
}

bool wgrad_ring_eligible(const WgP& p, int dtype) {
  static const bool no_ring = getenv("EVT_NO_RING") != nullptr;
  if (no_ring || dtype != EVT_DT_HALF) return false;
  if (p.CA % 64 || p.CB % 32) return false;
  if (p.Aact || p.Bact || p.a_slope != 1.f || p.b_slope != 1.f) return false;
  if (p.LA != p.Q) return false;
  const long units = (long)p.nseq * p.Q;
  if (units >= (1L << 31) - WPOS || units < 512) return false;
  return true;
}

int launch_wgrad_ring(const WgP& p_in, hipStream_t st) {
  WgP p = p_in;
  if (!wgrad_ring_eligible(p, EVT_DT_HALF)) return EVT_ENOTSUP;
  const int KT = p.KHp <= 1 ? 1 : (p.KHp <= 3 ? 3 : 5);
  p.nchunk = p.CB / 32;
  p.ntapgrp = (p.KHp + KT - 1) / KT;
  const int nstages = (int)(((long)p.nseq * p.Q + WPOS - 1) / WPOS);
  // 128-channel tiles when that still yields enough blocks, else 64
  const long tiles128 = p.CA % 128 == 0 ? (long)(p.CA / 128) * p.nchunk * p.ntapgrp : 0;
  const int MA = (tiles128 >= 128 || (tiles128 > 0 && tiles128 * (nstages / 4) >= 512)) ? 4 : 2;
  const long tiles = (long)(p.CA / (32 * MA)) * p.nchunk * p.ntapgrp;
  static const long target = getenv("EVT_RING_BLOCKS") ? atol(getenv("EVT_RING_BLOCKS")) : 256;   // tuning knob (measured: 256 best)
  int nsplit, per;                                // ~1 block per CU: more splits only add partial tiles; >= 3 K stages per block
  wgrad_pick_split(p, tiles, nstages, target, 3, &nsplit, &per);
  // XCD-aware order, EVT_WGRAD_RING_XCD=1 (default: plain grid): the split count rounded DOWN to a multiple of 8 (it is
  // bounded by the slabs the caller holds), every split non-empty.  Measured (round 4): FETCH_SIZE of the WN in-layer
  // gradient (192 -> 384 k5, 3200 positions, 36 tiles x 8 splits) 12.1 -> 1.9 MB per launch, and the launch 15 -> 20 us:
  // the 36 tiles of a split land on the 32 CUs of one XCD at once.  These operands fit every L2; time decides: off.
  static const bool xcd_on = getenv("EVT_WGRAD_RING_XCD") && atoi(getenv("EVT_WGRAD_RING_XCD")) == 1;
  p.xcd_order = 0;
  if (xcd_on && nsplit >= 8) {
    const int s8 = nsplit / 8 * 8;
    const int per8 = (nstages + s8 - 1) / s8;
    if ((long)per8 * (s8 - 1) < nstages) { nsplit = s8; per = per8; p.xcd_order = 1; }
  }
  p.nsplit = nsplit;
  p.now_used = p.prev_used > nsplit ? p.prev_used : nsplit;
  if (p.parts > 0 && p.used_host) *p.used_host = p.now_used;
#define RING(MA_, KT_) (MA_ == 4 && KT_ == 5 ? launch_ring_inst<MA_, KT_, 3>(p, per, st) : launch_ring_inst<MA_, KT_, 4>(p, per, st))
  if (MA == 4) return KT == 1 ? RING(4, 1) : (KT == 3 ? RING(4, 3) : RING(4, 5));
  return KT == 1 ? RING(2, 1) : (KT == 3 ? RING(2, 3) : RING(2, 5));
#undef RING
}

}  // namespace evt_conv
