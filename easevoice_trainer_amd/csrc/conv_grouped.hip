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
template <> struct GFrag<bf16_t> {
  static constexpr int EPL = 8, KS = 32;
  typedef bf16x8 type;
  static __device__ __forceinline__ f32x4 mma(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <typename T> union GBuf;
template <> union GBuf<float> { float v; float e[1]; };
template <> union GBuf<bf16_t> { bf16x8 v; bf16_t e[8]; };

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

// ---- forward: wave = (group, 64 output positions) --------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void grouped_fwd(GP p) {
  constexpr int EPL = GFrag<T>::EPL, KS = GFrag<T>::KS;
  typedef typename GFrag<T>::type frag_t;
  constexpr int R = 4 * (PT - 1) + KPAD / 4;  // staged rows per tile (300)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g8 = lane >> 4;
  T* wsA = reinterpret_cast<T*>(smem) + wave * (16 * WP);
  T* xsA = reinterpret_cast<T*>(smem) + 4 * (16 * WP) + wave * (R * 4 + 16);
  const int seq = blockIdx.x / p.tiles_per_seq;
  const int q0 = (blockIdx.x % p.tiles_per_seq) * PT;
  const int grp0 = blockIdx.y * 4;
  const int grp = grp0 + wave;
  const int K = p.k * 4;
  const T* xg = reinterpret_cast<const T*>(p.x) + (long)seq * p.lin * p.cin;
  // stage weights of this wave's group: [16 co][K] contiguous in the REG image, zero-padded to KPAD
  const T* wg = reinterpret_cast<const T*>(p.w) + (long)grp * p.cog * K;
  for (int idx = lane; idx < 16 * KPAD; idx += 64) {
    const int co = idx / KPAD, kk = idx - co * KPAD;
    wsA[co * WP + kk] = (kk < K && co < p.cog) ? wg[co * K + kk] : from_f<T>(0.f);
  }
  // stage rows 4*q0 - pad + [0, R) of the block's 4 groups (16 contiguous channels per row)
  T* xs_all = reinterpret_cast<T*>(smem) + 4 * (16 * WP);
  const int row0 = 4 * q0 - p.pad;
  for (int idx = tid; idx < R * 4; idx += 256) {
    const int r = idx >> 2, gl = idx & 3;  // row, local group
    const int row = row0 + r;
    T v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = from_f<T>(0.f);
    if (row >= 0 && row < p.lin) {
      const T* src = xg + (long)row * p.cin + (grp0 + gl) * 4;
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = from_f<T>(lrelu_f(to_f<T>(src[c]), p.in_slope));
    }
    T* dst = xs_all + gl * (R * 4 + 16) + r * 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) dst[c] = v[c];
  }
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
      float v = acc[j][r];
      if (p.bias) v += p.bias[co + r];
      if (p.out_act == EVT_ACT_LRELU) v = lrelu_f(v, p.out_slope);
      else if (p.out_act == EVT_ACT_TANH) v = tanhf(v);
      outv[r] = from_f<T>(v);
    }
    T* dst = yg + (long)q * p.cout + co;
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[r] = outv[r];
  }
}

// ---- backward-data: wave = (group, 64 values of q'), rows 4q' + phase - pad, 16 MFMA rows = (phase, c) -------
template <typename T>
__global__ __launch_bounds__(256) void grouped_bwd_data(GP p) {
  constexpr int EPL = GFrag<T>::EPL, KS = GFrag<T>::KS;
  typedef typename GFrag<T>::type frag_t;
  constexpr int JP = KPAD / 16;            // 12 padded "j" taps of 16 output channels
  constexpr int R = PT + JP - 1;           // staged dy rows per tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, g8 = lane >> 4;
  T* wsA = reinterpret_cast<T*>(smem) + wave * (16 * WP);
  T* dsA = reinterpret_cast<T*>(smem) + 4 * (16 * WP) + wave * (R * 16 + 16);
  const int seq = blockIdx.x / p.tiles_per_seq;
  const int q0 = (blockIdx.x % p.tiles_per_seq) * PT;   // first q' of the tile
  const int grp = blockIdx.y * 4 + wave;
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
  // stage dy rows q0 - (JP-1) + [0, R) of this wave's group, times act'(y)
  const T* dyg = reinterpret_cast<const T*>(p.x) + (long)seq * p.lout * p.cout;
  const T* yag = p.xact ? reinterpret_cast<const T*>(p.xact) + (long)seq * p.lout * p.cout : nullptr;
  const int r0 = q0 - (JP - 1);
  for (int idx = lane; idx < R * 16; idx += 64) {
    const int r = idx >> 4, co = idx & 15;
    const int row = r0 + r;
    T v = from_f<T>(0.f);
    if (row >= 0 && row < p.lout && co < p.cog) {
      const long off = (long)row * p.cout + grp * p.cog + co;
      float f = to_f<T>(dyg[off]);
      if (yag) f *= dact_from_out(p.out_act, to_f<T>(yag[off]), p.out_slope);
      v = from_f<T>(f);
    }
    dsA[r * 16 + co] = v;
  }
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
    T* dst = dxg + (long)row * p.cin + grp * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[r] = from_f<T>(acc[j][r]);
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

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

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
  const int sz = c->dtype == EVT_DT_BF16 ? 2 : 4;
  const size_t lds = (size_t)(4 * 16 * WP + 4 * ((4 * (PT - 1) + KPAD / 4) * 4 + 16)) * sz;
  dim3 grid(p.nseq * p.tiles_per_seq, c->groups / 4);
  hipStream_t st = (hipStream_t)stream;
  if (c->dtype == EVT_DT_BF16) {
    if (set_lds(&grouped_fwd<bf16_t>, lds)) return EVT_ELAUNCH;
    hipLaunchKernelGGL(grouped_fwd<bf16_t>, grid, dim3(256), lds, st, p);
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
  const int sz = c->dtype == EVT_DT_BF16 ? 2 : 4;
  const size_t lds = (size_t)(4 * 16 * WP + 4 * ((PT + KPAD / 16 - 1) * 16 + 16)) * sz;
  dim3 grid(p.nseq * p.tiles_per_seq, c->groups / 4);
  hipStream_t st = (hipStream_t)stream;
  if (c->dtype == EVT_DT_BF16) {
    if (set_lds(&grouped_bwd_data<bf16_t>, lds)) return EVT_ELAUNCH;
    hipLaunchKernelGGL(grouped_bwd_data<bf16_t>, grid, dim3(256), lds, st, p);
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
  const int sz = c->dtype == EVT_DT_BF16 ? 2 : 4;
  const size_t lds = (size_t)4 * (PT * 16 + (4 * (PT - 1) + 44) * 4 + 32) * sz;
  dim3 grid(p.nsplit, c->groups / 4);
  hipStream_t st = (hipStream_t)stream;
  if (c->dtype == EVT_DT_BF16) {
    if (set_lds(&grouped_bwd_weight<bf16_t>, lds)) return EVT_ELAUNCH;
    hipLaunchKernelGGL(grouped_bwd_weight<bf16_t>, grid, dim3(256), lds, st, p);
  } else {
    if (set_lds(&grouped_bwd_weight<float>, lds)) return EVT_ELAUNCH;
    hipLaunchKernelGGL(grouped_bwd_weight<float>, grid, dim3(256), lds, st, p);
  }
  return evt_check_launch();
}
