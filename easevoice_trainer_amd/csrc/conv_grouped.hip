// Grouped strided Conv1d of DiscriminatorS (src/easevoice/module/models.py:567-570: k=41, stride 4, 4 input and
// 16 output channels per group) on MFMA, forward / backward-data / backward-weight.  gfx950 only.
//
// With cin_per_group (4) * stride (4) == 16, the im2col matrix of ONE group is a strided view of that group's
// channels-last rows laid flat in LDS:  B[k = tap*4 + c][n = q] = xs[16*q + k].  So a wave owns one group and a
// tile of positions, stages the group's rows once, and every MFMA operand is an aligned 16-byte LDS read.
// K = 41*4 = 164 is zero-padded to 192 (six 32-deep bf16 steps / 48 fp32 steps).
// Backward-data uses the same trick on dy: the 4 output phases x 4 channels form the 16 MFMA rows, and
// B[k = j*16 + co][n = q'] = dys[16*q' + k] over the 16 output channels of the group.
#include "evt_common.h"
#include "../../include/evt.h"

namespace {

template <typename T> struct GFrag;
template <> struct GFrag<float> {
  static constexpr int EPL = 1, KS = 4;
  typedef float type;
  static __device__ __forceinline__ f32x4 mma(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
};
template <> struct GFrag<h16_t> {
  static constexpr int EPL = 8, KS = 32;
  typedef h16x8 type;
  static __device__ __forceinline__ f32x4 mma(h16x8 a, h16x8 b, f32x4 c) {
    return EVT_MFMA_16x16x32(a, b, c, 0, 0, 0);
  }
};
template <typename T> union GBuf;
template <> union GBuf<float> { float v; float e[1]; };
template <> union GBuf<h16_t> { h16x8 v; h16_t e[8]; };

constexpr int KPAD = 192;   // padded K (taps*4 or taps*16)
constexpr int WP = KPAD + 8;  // weight row pitch in elements (16-byte aligned, odd multiple of 16 B)
constexpr int PT = 64;      // positions per wave tile

struct GP {
  const void* x;     // fwd: x [nseq][lin][cin] ; bwd-data: dy [nseq][lout][cout]
  const void* xact;  // bwd-data: saved y (activation output) or null
  const void* w;     // REG image [cout][k][4]
  const float* bias;
  void* y;           // fwd: y ; bwd-data: dx
  float* dw;         // bwd-weight
  const void* dy;    // bwd-weight
  int nseq, lin, lout, cin, cout, k, pad, groups;
  float in_slope;
  int out_act;
  float out_slope;
  int tiles_per_seq;
  int nsplit;
  int cog;           // output channels per group: 16, or 4 (MFMA rows 4..15 are zero padding)
};

// 16-byte staging of channels-last rows of the block's 4 groups (16 channels = 32 B bf16 / 64 B fp32 per row) into the
// per-group flat arrays gs[g][r*4 + c]; rows outside [0, nrows_total) are zeros; optional leaky-relu on load.
template <typename T>
__device__ __forceinline__ void stage_group_rows(T* gs_all, int garr, const T* src_seq, int ld, int ch0, int row0, int R,
                                                 int nrows_total, float slope) {
  constexpr int V = 16 / sizeof(T);          // channels per 16-byte piece: 8 (2 groups) or 4 (1 group)
  constexpr int PPR = 16 / V;                // pieces per row
  for (int idx = threadIdx.x; idx < R * PPR; idx += 256) {
    const int r = idx / PPR, part = idx - r * PPR;
    const int row = row0 + r;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row >= 0 && row < nrows_total) {
      v = *reinterpret_cast<const uint4*>(src_seq + (long)row * ld + ch0 + part * V);
      if (slope != 1.f) {
        T* h = reinterpret_cast<T*>(&v);
#pragma unroll
        for (int e = 0; e < V; ++e) h[e] = from_f<T>(lrelu_f(to_f<T>(h[e]), slope));
      }
    }
    const uint2* h2 = reinterpret_cast<const uint2*>(&v);
    if constexpr (sizeof(T) == 2) {
      *reinterpret_cast<uint2*>(gs_all + (2 * part) * garr + r * 4) = h2[0];
      *reinterpret_cast<uint2*>(gs_all + (2 * part + 1) * garr + r * 4) = h2[1];
    } else {
      *reinterpret_cast<uint4*>(gs_all + part * garr + r * 4) = v;
    }
  }
}

// ---- forward: wave = one group; the block (4 groups) walks position tiles of 64 with its weights staged ONCE ----
template <typename T>
__global__ __launch_bounds__(256) void grouped_fwd(GP p) {
  constexpr int EPL = GFrag<T>::EPL, KS = GFrag<T>::KS;
  typedef typename GFrag<T>::type frag_t;
  constexpr int R = 4 * (PT - 1) + KPAD / 4;  // staged rows per tile (300)
  constexpr int GARR = R * 4 + 16;            // per-group flat array (elements)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g8 = lane >> 4;
  T* wsA = reinterpret_cast<T*>(smem) + wave * (16 * WP);
  T* xs_all = reinterpret_cast<T*>(smem) + 4 * (16 * WP);
  T* xsA = xs_all + wave * GARR;
  const int grp0 = blockIdx.y * 4;
  const int grp = grp0 + wave;
  const int K = p.k * 4;
  // stage weights of this wave's group: [16 co][K] contiguous in the REG image, zero-padded to KPAD
  const T* wg = reinterpret_cast<const T*>(p.w) + (long)grp * p.cog * K;
  for (int idx = lane; idx < 16 * KPAD; idx += 64) {
    const int co = idx / KPAD, kk = idx - co * KPAD;
    wsA[co * WP + kk] = (kk < K && co < p.cog) ? wg[co * K + kk] : from_f<T>(0.f);
  }
  float bv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bv[r] = (p.bias && g8 * 4 < p.cog) ? p.bias[grp * p.cog + g8 * 4 + r] : 0.f;
  const long total = (long)p.nseq * p.tiles_per_seq;
  for (long tile = blockIdx.x; tile < total; tile += gridDim.x) {
    const int seq = (int)(tile / p.tiles_per_seq);
    const int q0 = (int)(tile - (long)seq * p.tiles_per_seq) * PT;
    const T* xg = reinterpret_cast<const T*>(p.x) + (long)seq * p.lin * p.cin;
    __syncthreads();     // previous tile's fragment reads are done (also orders the weight staging before first use)
    stage_group_rows<T>(xs_all, GARR, xg, p.cin, grp0 * 4, 4 * q0 - p.pad, R, p.lin, p.in_slope);
    __syncthreads();
    f32x4 acc[PT / 16];
#pragma unroll
    for (int j = 0; j < PT / 16; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int st = 0; st < KPAD / KS; ++st) {
      const int kl = st * KS + g8 * EPL;
      const frag_t a = *reinterpret_cast<const frag_t*>(wsA + n * WP + kl);
#pragma unroll
      for (int j = 0; j < PT / 16; ++j) {
        const frag_t b = *reinterpret_cast<const frag_t*>(xsA + 16 * (j * 16 + n) + kl);
        acc[j] = GFrag<T>::mma(a, b, acc[j]);
      }
    }
    T* yg = reinterpret_cast<T*>(p.y) + (long)seq * p.lout * p.cout;
#pragma unroll
    for (int j = 0; j < PT / 16; ++j) {
      const int q = q0 + j * 16 + n;
      if (q >= p.lout || g8 * 4 >= p.cog) continue;
      const int co = grp * p.cog + g8 * 4;
      T outv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[j][r] + bv[r];
        if (p.out_act == EVT_ACT_LRELU) v = lrelu_f(v, p.out_slope);
        else if (p.out_act == EVT_ACT_TANH) v = tanhf(v);
        outv[r] = from_f<T>(v);
      }
      T* dst = yg + (long)q * p.cout + co;
      if constexpr (sizeof(T) == 2) *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<uint2*>(outv);
      else *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<uint4*>(outv);
    }
  }
}

// 16-byte staging of dy rows (the block's 4 groups = 4*cog contiguous channels) into per-group arrays gs[g][r*16 + co],
// times act'(y) when `ya` is given; slots co >= cog stay zero (the arrays are cleared once per block).
template <typename T>
__device__ __forceinline__ void stage_dy_rows(T* gs_all, int garr, const T* dy_seq, const T* ya_seq, int ld, int ch0, int cog,
                                              int row0, int R, int nrows_total, int act, float slope) {
  constexpr int V = 16 / sizeof(T);
  const int ppr = 4 * cog / V;               // pieces per row (cog in {4, 16}: 2/8 bf16, 4/16 fp32)
  for (int idx = threadIdx.x; idx < R * ppr; idx += 256) {
    const int r = idx / ppr, part = idx - r * ppr;
    const int row = row0 + r;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row >= 0 && row < nrows_total) {
      const long off = (long)row * ld + ch0 + part * V;
      v = *reinterpret_cast<const uint4*>(dy_seq + off);
      if (ya_seq) {
        const uint4 va = *reinterpret_cast<const uint4*>(ya_seq + off);
        T* h = reinterpret_cast<T*>(&v);
        const T* ha = reinterpret_cast<const T*>(&va);
#pragma unroll
        for (int e = 0; e < V; ++e) h[e] = from_f<T>(to_f<T>(h[e]) * dact_from_out(act, to_f<T>(ha[e]), slope));
      }
    }
    const int c = part * V;                  // first channel of the piece inside the block's 4*cog channels
    if (cog >= V) {
      *reinterpret_cast<uint4*>(gs_all + (c / cog) * garr + r * 16 + (c % cog)) = v;
    } else {                                 // bf16, cog = 4: the piece holds two groups of 4 channels
      const uint2* h2 = reinterpret_cast<const uint2*>(&v);
      *reinterpret_cast<uint2*>(gs_all + (c / cog) * garr + r * 16) = h2[0];
      *reinterpret_cast<uint2*>(gs_all + (c / cog + 1) * garr + r * 16) = h2[1];
    }
  }
}

// ---- backward-data: wave = one group, rows 4q' + phase - pad, 16 MFMA rows = (phase, c); persistent over q' tiles ----
template <typename T>
__global__ __launch_bounds__(256) void grouped_bwd_data(GP p) {
  constexpr int EPL = GFrag<T>::EPL, KS = GFrag<T>::KS;
  typedef typename GFrag<T>::type frag_t;
  constexpr int JP = KPAD / 16;            // 12 padded "j" taps of 16 output channels
  constexpr int R = PT + JP - 1;           // staged dy rows per tile
  constexpr int GARR = R * 16 + 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g8 = lane >> 4;
  T* wsA = reinterpret_cast<T*>(smem) + wave * (16 * WP);
  T* ds_all = reinterpret_cast<T*>(smem) + 4 * (16 * WP);
  T* dsA = ds_all + wave * GARR;
  const int grp0 = blockIdx.y * 4;
  const int grp = grp0 + wave;
  const int K = p.k * 4;
  // A[(phase, c)][(jj, co)] = w[grp*16+co][t = phase + 4*(JP-1-jj)][c], zero when t >= k
  const T* wg = reinterpret_cast<const T*>(p.w) + (long)grp * p.cog * K;
  for (int idx = lane; idx < 16 * KPAD; idx += 64) {
    const int m = idx / KPAD, kk = idx - m * KPAD;
    const int ph = m >> 2, c = m & 3;
    const int jj = kk >> 4, co = kk & 15;
    const int t = ph + 4 * (JP - 1 - jj);
    wsA[m * WP + kk] = (t < p.k && co < p.cog) ? wg[co * K + t * 4 + c] : from_f<T>(0.f);
  }
  for (int idx = tid; idx < 4 * GARR; idx += 256) ds_all[idx] = from_f<T>(0.f);
  const long total = (long)p.nseq * p.tiles_per_seq;
  for (long tile = blockIdx.x; tile < total; tile += gridDim.x) {
    const int seq = (int)(tile / p.tiles_per_seq);
    const int q0 = (int)(tile - (long)seq * p.tiles_per_seq) * PT;   // first q' of the tile
    const T* dyg = reinterpret_cast<const T*>(p.x) + (long)seq * p.lout * p.cout;
    const T* yag = p.xact ? reinterpret_cast<const T*>(p.xact) + (long)seq * p.lout * p.cout : nullptr;
    __syncthreads();
    stage_dy_rows<T>(ds_all, GARR, dyg, yag, p.cout, grp0 * p.cog, p.cog, q0 - (JP - 1), R, p.lout, p.out_act, p.out_slope);
    __syncthreads();
    f32x4 acc[PT / 16];
#pragma unroll
    for (int j = 0; j < PT / 16; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int st = 0; st < KPAD / KS; ++st) {
      const int kl = st * KS + g8 * EPL;
      const frag_t a = *reinterpret_cast<const frag_t*>(wsA + n * WP + kl);
#pragma unroll
      for (int j = 0; j < PT / 16; ++j) {
        const frag_t b = *reinterpret_cast<const frag_t*>(dsA + 16 * (j * 16 + n) + kl);
        acc[j] = GFrag<T>::mma(a, b, acc[j]);
      }
    }
    // lane holds MFMA rows g8*4 + r = (phase g8, channel r) of column q'
    T* dxg = reinterpret_cast<T*>(p.y) + (long)seq * p.lin * p.cin;
#pragma unroll
    for (int j = 0; j < PT / 16; ++j) {
      const int qp = q0 + j * 16 + n;
      const int row = 4 * qp + g8 - p.pad;
      if (row < 0 || row >= p.lin) continue;
      T outv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) outv[r] = from_f<T>(acc[j][r]);
      T* dst = dxg + (long)row * p.cin + grp * 4;
      if constexpr (sizeof(T) == 2) *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<uint2*>(outv);
      else *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<uint4*>(outv);
    }
  }
}

// ---- backward-weight: wave = group; blocks split the positions; dW[co][t*4+c] += dy[q][co] * x[16q + t*4 + c] ----
template <typename T>
__global__ __launch_bounds__(256) void grouped_bwd_weight(GP p) {
  constexpr int EPL = GFrag<T>::EPL, KS = GFrag<T>::KS;
  typedef typename GFrag<T>::type frag_t;
  constexpr int NTB = 11;                     // 11 tiles of 16 cover K = 164 (176)
  constexpr int R = 4 * (PT - 1) + 44;        // rows touched by 64 positions x 44 (padded) taps
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g8 = lane >> 4;
  T* dsA = reinterpret_cast<T*>(smem) + wave * (PT * 16 + R * 4 + 32);
  T* xsA = dsA + PT * 16 + 16;
  const int grp = blockIdx.y * 4 + wave;
  const int K = p.k * 4;
  const T* xg0 = reinterpret_cast<const T*>(p.x);
  const T* dy0 = reinterpret_cast<const T*>(p.dy);
  const T* ya0 = reinterpret_cast<const T*>(p.xact);
  f32x4 acc[NTB];
#pragma unroll
  for (int j = 0; j < NTB; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long total = (long)p.nseq * p.tiles_per_seq;
  for (long it = blockIdx.x; it < total; it += p.nsplit) {
    const int seq = (int)(it / p.tiles_per_seq);
    const int q0 = (int)(it % p.tiles_per_seq) * PT;
    __syncthreads();
    for (int idx = lane; idx < PT * 16; idx += 64) {
      const int r = idx >> 4, co = idx & 15;
      const int q = q0 + r;
      T v = from_f<T>(0.f);
      if (q < p.lout && co < p.cog) {
        const long off = ((long)seq * p.lout + q) * p.cout + grp * p.cog + co;
        float f = to_f<T>(dy0[off]);
        if (ya0) f *= dact_from_out(p.out_act, to_f<T>(ya0[off]), p.out_slope);
        v = from_f<T>(f);
      }
      dsA[r * 16 + co] = v;
    }
    const int row0 = 4 * q0 - p.pad;
    for (int idx = lane; idx < R * 4; idx += 64) {
      const int r = idx >> 2, c = idx & 3;
      const int row = row0 + r;
      T v = from_f<T>(0.f);
      if (row >= 0 && row < p.lin)
        v = from_f<T>(lrelu_f(to_f<T>(xg0[((long)seq * p.lin + row) * p.cin + grp * 4 + c]), p.in_slope));
      xsA[r * 4 + c] = v;
    }
    __syncthreads();
#pragma unroll 1
    for (int kk = 0; kk < PT / KS; ++kk) {
      const int k0 = kk * KS + g8 * EPL;
      GBuf<T> av;
#pragma unroll
      for (int e = 0; e < EPL; ++e) av.e[e] = dsA[(k0 + e) * 16 + n];
      const frag_t a = av.v;
#pragma unroll
      for (int j = 0; j < NTB; ++j) {
        GBuf<T> bv;
#pragma unroll
        for (int e = 0; e < EPL; ++e) bv.e[e] = xsA[16 * (k0 + e) + j * 16 + n];
        acc[j] = GFrag<T>::mma(a, bv.v, acc[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NTB; ++j) {
    const int kidx = j * 16 + n;
    if (kidx >= K) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (g8 * 4 + r >= p.cog) continue;
      const int co = grp * p.cog + g8 * 4 + r;
      atomicAdd(p.dw + (long)co * K + kidx, acc[j][r]);
    }
  }
}

// ---- backward-weight, bf16: vector staging + transpose-read fragments -------------------------------------------
// Same GEMM as above (M = the group's 16 output channels, N = (tap, c) = 11 tiles of 16, K = positions).  dy is
// staged position-major for the block's 4 groups, x as the flat per-group row array of the forward kernel, and BOTH
// MFMA operands -- 8 consecutive positions of one column -- come from ds_read_b64_tr_b16: the row address is per lane,
// so the overlapping-window view B[pos][kidx] = xs[16*pos + kidx] is just an address (the gather version issued 8
// two-byte LDS reads per fragment and 2-byte global loads to stage).
__device__ __forceinline__ h16x8 g_tr2(const h16_t* p0, const h16_t* p1) {
  const unsigned a0 = (unsigned)(uintptr_t)p0, a1 = (unsigned)(uintptr_t)p1;
  uint2 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(lo), "=&v"(hi) : "v"(a0), "v"(a1) : "memory");
  union { uint4 u; h16x8 v; } r;
  r.u = make_uint4(lo.x, lo.y, hi.x, hi.y);
  return r.v;
}

// four fragments (eight transpose reads) behind ONE wait
__device__ __forceinline__ void g_tr2x4(const h16_t* p0, const h16_t* p1, const h16_t* p2, const h16_t* p3, int hi_off,
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

__global__ __launch_bounds__(256) void grouped_bwd_weight_tr(GP p) {
  typedef h16_t T;
  constexpr int NTB = 11;                     // 11 tiles of 16 cover K = 164 (176)
  constexpr int R = 4 * (PT - 1) + 44;        // rows touched by 64 positions x 44 (padded) taps
  constexpr int GARR = R * 4 + 32;            // flat per-group x array (+ slack for the padded taps of the last rows)
  constexpr int DARR = PT * 16;               // per-group dy tile [pos][16]
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g8 = lane >> 4;
  T* ds_all = reinterpret_cast<T*>(smem);
  T* xs_all = ds_all + 4 * DARR;
  const T* dsA = ds_all + wave * DARR;
  const T* xsA = xs_all + wave * GARR;
  const int grp0 = blockIdx.y * 4, grp = grp0 + wave;
  const int K = p.k * 4;
  for (int idx = tid; idx < 4 * DARR + 4 * GARR; idx += 256) ds_all[idx] = (T)0;
  f32x4 acc[NTB];
#pragma unroll
  for (int j = 0; j < NTB; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long total = (long)p.nseq * p.tiles_per_seq;
  for (long it = blockIdx.x; it < total; it += p.nsplit) {
    const int seq = (int)(it / p.tiles_per_seq);
    const int q0 = (int)(it % p.tiles_per_seq) * PT;
    const T* dyg = reinterpret_cast<const T*>(p.dy) + (long)seq * p.lout * p.cout;
    const T* yag = p.xact ? reinterpret_cast<const T*>(p.xact) + (long)seq * p.lout * p.cout : nullptr;
    const T* xg = reinterpret_cast<const T*>(p.x) + (long)seq * p.lin * p.cin;
    __syncthreads();
    stage_dy_rows<T>(ds_all, DARR, dyg, yag, p.cout, grp0 * p.cog, p.cog, q0, PT, p.lout, p.out_act, p.out_slope);
    stage_group_rows<T>(xs_all, GARR, xg, p.cin, grp0 * 4, 4 * q0 - p.pad, R, p.lin, p.in_slope);
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < PT / 32; ++kk) {
      const int kb = kk * 32 + g8 * 8;        // first of this lane group's 8 positions
      const T* pa = dsA + (kb + (n >> 2)) * 16 + 4 * (n & 3);
      const h16x8 a = g_tr2(pa, pa + 4 * 16);
      const T* pb = xsA + 16 * (kb + (n >> 2)) + 4 * (n & 3);
#pragma unroll
      for (int j0 = 0; j0 < 8; j0 += 4) {
        h16x8 b0, b1, b2, b3;
        g_tr2x4(pb + j0 * 16, pb + (j0 + 1) * 16, pb + (j0 + 2) * 16, pb + (j0 + 3) * 16, 4 * 16 * 2, b0, b1, b2, b3);
        acc[j0] = EVT_MFMA_16x16x32(a, b0, acc[j0], 0, 0, 0);
        acc[j0 + 1] = EVT_MFMA_16x16x32(a, b1, acc[j0 + 1], 0, 0, 0);
        acc[j0 + 2] = EVT_MFMA_16x16x32(a, b2, acc[j0 + 2], 0, 0, 0);
        acc[j0 + 3] = EVT_MFMA_16x16x32(a, b3, acc[j0 + 3], 0, 0, 0);
      }
      {   // tiles 8..10 (+ a re-read of tile 10 to fill the batch)
        h16x8 b0, b1, b2, b3;
        g_tr2x4(pb + 8 * 16, pb + 9 * 16, pb + 10 * 16, pb + 10 * 16, 4 * 16 * 2, b0, b1, b2, b3);
        acc[8] = EVT_MFMA_16x16x32(a, b0, acc[8], 0, 0, 0);
        acc[9] = EVT_MFMA_16x16x32(a, b1, acc[9], 0, 0, 0);
        acc[10] = EVT_MFMA_16x16x32(a, b2, acc[10], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NTB; ++j) {
    const int kidx = j * 16 + n;
    if (kidx >= K) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (g8 * 4 + r >= p.cog) continue;
      const int co = grp * p.cog + g8 * 4 + r;
      atomicAdd(p.dw + (long)co * K + kidx, acc[j][r]);
    }
  }
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// blocks along x for the persistent forward / backward-data kernels: ~3 blocks per CU over all group sets, each block
// amortises its weight staging (3072 elements per wave) over several position tiles
inline int persistent_blocks(long tiles, int group_sets) {
  long b = (768 + group_sets - 1) / group_sets;
  if (b > tiles) b = tiles;
  if (b < 1) b = 1;
  return (int)b;
}

template <typename F>
int set_lds(F f, size_t lds) {
  if (lds > 48 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess)
      return EVT_ELAUNCH;
  }
  return EVT_OK;
}

}  // namespace

// Shape gate shared with conv1d.hip's dispatcher.
extern "C" int evt_grouped_supported(const evt_conv1d_params* c) {
  return !c->transposed && c->groups > 1 && c->groups % 4 == 0 && c->cin / c->groups == 4 && (c->cout / c->groups == 16 || c->cout / c->groups == 4) &&
         c->stride == 4 && c->dil == 1 && c->k * 4 <= 176 && c->k >= 4;
}

extern "C" int evt_grouped_fwd(const evt_conv1d_params* c, const void* x, const void* w_reg, const float* bias, void* y,
                               void* stream) {
  GP p{};
  p.x = x; p.w = w_reg; p.bias = bias; p.y = y;
  p.nseq = c->nseq; p.lin = c->lin; p.lout = evt_conv1d_lout(c); p.cin = c->cin; p.cout = c->cout; p.k = c->k;
  p.pad = c->pad; p.groups = c->groups; p.in_slope = c->in_slope; p.out_act = c->out_act; p.out_slope = c->out_slope;
  p.cog = c->cout / c->groups;
  p.tiles_per_seq = cdiv(p.lout, PT);
  const int sz = c->dtype == EVT_DT_HALF ? 2 : 4;
  const size_t lds = (size_t)(4 * 16 * WP + 4 * ((4 * (PT - 1) + KPAD / 4) * 4 + 16)) * sz;
  dim3 grid(persistent_blocks((long)p.nseq * p.tiles_per_seq, c->groups / 4), c->groups / 4);
  hipStream_t st = (hipStream_t)stream;
  if (c->dtype == EVT_DT_HALF) {
    if (set_lds(&grouped_fwd<h16_t>, lds)) return EVT_ELAUNCH;
    hipLaunchKernelGGL(grouped_fwd<h16_t>, grid, dim3(256), lds, st, p);
  } else {
    if (set_lds(&grouped_fwd<float>, lds)) return EVT_ELAUNCH;
    hipLaunchKernelGGL(grouped_fwd<float>, grid, dim3(256), lds, st, p);
  }
  return evt_check_launch();
}

extern "C" int evt_grouped_bwd_data(const evt_conv1d_params* c, const void* dy, const void* y, const void* w_reg,
                                    void* dx, void* stream) {
  GP p{};
  p.x = dy; p.xact = c->out_act != EVT_ACT_NONE ? y : nullptr; p.w = w_reg; p.y = dx;
  p.nseq = c->nseq; p.lin = c->lin; p.lout = evt_conv1d_lout(c); p.cin = c->cin; p.cout = c->cout; p.k = c->k;
  p.pad = c->pad; p.groups = c->groups; p.in_slope = c->in_slope; p.out_act = c->out_act; p.out_slope = c->out_slope;
  p.cog = c->cout / c->groups;
  const int nq = (c->lin - 1 + c->pad) / 4 + 1;   // q' in [0, nq)
  p.tiles_per_seq = cdiv(nq, PT);
  const int sz = c->dtype == EVT_DT_HALF ? 2 : 4;
  const size_t lds = (size_t)(4 * 16 * WP + 4 * ((PT + KPAD / 16 - 1) * 16 + 16)) * sz;
  dim3 grid(persistent_blocks((long)p.nseq * p.tiles_per_seq, c->groups / 4), c->groups / 4);
  hipStream_t st = (hipStream_t)stream;
  if (c->dtype == EVT_DT_HALF) {
    if (set_lds(&grouped_bwd_data<h16_t>, lds)) return EVT_ELAUNCH;
    hipLaunchKernelGGL(grouped_bwd_data<h16_t>, grid, dim3(256), lds, st, p);
  } else {
    if (set_lds(&grouped_bwd_data<float>, lds)) return EVT_ELAUNCH;
    hipLaunchKernelGGL(grouped_bwd_data<float>, grid, dim3(256), lds, st, p);
  }
  return evt_check_launch();
}

extern "C" int evt_grouped_bwd_weight(const evt_conv1d_params* c, const void* x, const void* dy, const void* y, float* dw,
                                      void* stream) {
  GP p{};
  p.x = x; p.dy = dy; p.xact = c->out_act != EVT_ACT_NONE ? y : nullptr; p.dw = dw;
  p.nseq = c->nseq; p.lin = c->lin; p.lout = evt_conv1d_lout(c); p.cin = c->cin; p.cout = c->cout; p.k = c->k;
  p.pad = c->pad; p.groups = c->groups; p.in_slope = c->in_slope; p.out_act = c->out_act; p.out_slope = c->out_slope;
  p.cog = c->cout / c->groups;
  p.tiles_per_seq = cdiv(p.lout, PT);
  const long total = (long)p.nseq * p.tiles_per_seq;
  long split = 1024 / (c->groups / 4);
  if (split > 256) split = 256;
  if (split < 16) split = 16;
  if (split > total) split = total;
  if (split < 1) split = 1;
  p.nsplit = (int)split;
  const int sz = c->dtype == EVT_DT_HALF ? 2 : 4;
  const size_t lds = (size_t)4 * (PT * 16 + (4 * (PT - 1) + 44) * 4 + 32) * sz;
  dim3 grid(p.nsplit, c->groups / 4);
  hipStream_t st = (hipStream_t)stream;
  if (c->dtype == EVT_DT_HALF) {
    const size_t lds_tr = (size_t)(4 * PT * 16 + 4 * ((4 * (PT - 1) + 44) * 4 + 32)) * 2;
    if (set_lds(&grouped_bwd_weight_tr, lds_tr)) return EVT_ELAUNCH;
    hipLaunchKernelGGL(grouped_bwd_weight_tr, grid, dim3(256), lds_tr, st, p);
  } else {
    if (set_lds(&grouped_bwd_weight<float>, lds)) return EVT_ELAUNCH;
    hipLaunchKernelGGL(grouped_bwd_weight<float>, grid, dim3(256), lds, st, p);
  }
  return evt_check_launch();
}
