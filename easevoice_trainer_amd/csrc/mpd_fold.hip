// Input side of the multi-period / multi-scale discriminators as ONE launch per waveform batch (gfx950).
//
// Reference: DiscriminatorP.forward (src/easevoice/module/models.py:541-547: reflect-pad the [B, 1, T] waveform to a
// multiple of the period p, view it as [B, 1, T/p, p]) and DiscriminatorS.forward (models.py:576-587: the waveform as it
// is), for the periods 2, 3, 5, 7, 11 of MultiPeriodDiscriminator (models.py:590-614).  On channels-last rows the
// Conv2d((k,1)) over [T/p, p] is a Conv1d over each of the B*p interleaved sequences, so sub-discriminator i wants
//     out_i[(b*p + ph)][j][0] = x[b][reflect(j*p + ph)]        reflect(t) = t < T ? t : 2T - 2 - t
// Through torch that is pad + view/transpose/reshape copy + dtype cast per sub-discriminator and call (~45 launches per s2
// step forward, ~20 backward: slice / view / cast gradients and the sum over the six consumers).  Here: one gather kernel
// writes all six prepared batches -- optionally of TWO sources stacked as [src0 ; src1] (real and generated audio) --
// and one kernel sums the six gradients back into d x.  Pure index arithmetic on a 2.6 MB waveform batch: launch-bound.
#include "evt_common.h"
#include "../../include/evt.h"

namespace {

constexpr int MAXP = 8;

struct FoldP {
  const void* src0; const void* src1;      // [n0][T], [n1][T]
  void* out[MAXP];                         // out[i]: [(n0+n1)*p_i][L_i][1], L_i = ceil(T / p_i)
  long start[MAXP + 1];                    // prefix sums of the outputs' element counts
  int period[MAXP];
  int nper, n0, n1, T;
  int bf0, bf1;                            // source i holds bf16 (else fp32): real audio arrives fp32, generated in the compute dtype
};

template <typename TO>
__global__ __launch_bounds__(256) void mpd_fold_kernel(FoldP p) {
  const long total = p.start[p.nper];
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    int i = 0;
#pragma unroll
    for (int k = 1; k < MAXP; ++k)
      if (k < p.nper && e >= p.start[k]) i = k;
    const int per = p.period[i];
    const int Li = (p.T + per - 1) / per;
    const long r = e - p.start[i];
    const int j = (int)(r % Li);
    const long row = r / Li;                 // b * per + ph
    const int ph = (int)(row % per);
    const int b = (int)(row / per);
    int t = j * per + ph;
    if (t >= p.T) t = 2 * p.T - 2 - t;
    const bool second = b >= p.n0;
    const void* s = second ? p.src1 : p.src0;
    const long at = (long)(second ? b - p.n0 : b) * p.T + t;
    const float v = (second ? p.bf1 : p.bf0) ? h2f(reinterpret_cast<const h16_t*>(s)[at])
                                             : reinterpret_cast<const float*>(s)[at];
    reinterpret_cast<TO*>(p.out[i])[r] = from_f<TO>(v);
  }
}

struct UnfoldP {
  const void* dout[MAXP];                  // gradients of the prepared batches, rows [row0_i, row0_i + n*p_i) of each
  int period[MAXP];
  int nper, n, T, b0;                      // b0: first item of the batch the gradient is taken for (n1 half: b0 = n0)
  void* dsrc;                              // [n][T]
};

template <typename TG, typename TO>
__global__ __launch_bounds__(256) void mpd_unfold_kernel(UnfoldP p) {
  const long total = (long)p.n * p.T;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int t = (int)(e % p.T);
    const int b = (int)(e / p.T);
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
      if (i >= p.nper) break;
      const int per = p.period[i];
      const int Li = (p.T + per - 1) / per;
      const TG* d = reinterpret_cast<const TG*>(p.dout[i]) + (long)(p.b0 + b) * per * Li;
      acc += to_f<TG>(d[(long)(t % per) * Li + t / per]);
      const int tr = 2 * p.T - 2 - t;        // the padded position that mirrors onto t (if it exists in this period's pad)
      if (tr >= p.T && tr < Li * per) acc += to_f<TG>(d[(long)(tr % per) * Li + tr / per]);
    }
    reinterpret_cast<TO*>(p.dsrc)[e] = from_f<TO>(acc);
  }
}

inline int grid_for(long n) {
  long b = (n + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" {

int evt_mpd_fold(int32_t src0_dtype, const void* src0, int32_t n0, int32_t src1_dtype, const void* src1, int32_t n1, int32_t T,
                 const int32_t* periods, int32_t nper, void* const* outs, int32_t out_dtype, void* stream) {
  if (!src0 || n0 <= 0 || n1 < 0 || (n1 > 0 && !src1) || T < 2 || !periods || !outs || nper <= 0 || nper > MAXP)
    return EVT_EINVAL;
  if ((src0_dtype != EVT_DT_F32 && src0_dtype != EVT_DT_HALF) || (n1 > 0 && src1_dtype != EVT_DT_F32 && src1_dtype != EVT_DT_HALF))
    return EVT_EINVAL;
  FoldP p{};
  p.src0 = src0; p.src1 = src1; p.n0 = n0; p.n1 = n1; p.T = T; p.nper = nper;
  p.bf0 = src0_dtype == EVT_DT_HALF; p.bf1 = src1_dtype == EVT_DT_HALF;
  long at = 0;
  for (int i = 0; i < nper; ++i) {
    if (periods[i] < 1 || periods[i] >= T || !outs[i]) return EVT_EINVAL;
    p.period[i] = periods[i];
    p.out[i] = outs[i];
    p.start[i] = at;
    at += (long)(n0 + n1) * periods[i] * ((T + periods[i] - 1) / periods[i]);
  }
  p.start[nper] = at;
  const dim3 g(grid_for(at)), b(256);
  hipStream_t st = (hipStream_t)stream;
  if (out_dtype == EVT_DT_F32) hipLaunchKernelGGL((mpd_fold_kernel<float>), g, b, 0, st, p);
  else if (out_dtype == EVT_DT_HALF) hipLaunchKernelGGL((mpd_fold_kernel<h16_t>), g, b, 0, st, p);
  else return EVT_EINVAL;
  return evt_check_launch();
}

int evt_mpd_unfold(int32_t grad_dtype, const void* const* douts, const int32_t* periods, int32_t nper, int32_t b0, int32_t n,
                   int32_t T, int32_t dsrc_dtype, void* dsrc, void* stream) {
  if (!douts || !periods || nper <= 0 || nper > MAXP || b0 < 0 || n <= 0 || T < 2 || !dsrc) return EVT_EINVAL;
  UnfoldP p{};
  p.nper = nper; p.n = n; p.T = T; p.b0 = b0; p.dsrc = dsrc;
  for (int i = 0; i < nper; ++i) {
    if (periods[i] < 1 || periods[i] >= T || !douts[i]) return EVT_EINVAL;
    p.period[i] = periods[i];
    p.dout[i] = douts[i];
  }
  const dim3 g(grid_for((long)n * T)), b(256);
  hipStream_t st = (hipStream_t)stream;
  if (grad_dtype == EVT_DT_F32 && dsrc_dtype == EVT_DT_F32) hipLaunchKernelGGL((mpd_unfold_kernel<float, float>), g, b, 0, st, p);
  else if (grad_dtype == EVT_DT_HALF && dsrc_dtype == EVT_DT_HALF) hipLaunchKernelGGL((mpd_unfold_kernel<h16_t, h16_t>), g, b, 0, st, p);
  else if (grad_dtype == EVT_DT_HALF && dsrc_dtype == EVT_DT_F32) hipLaunchKernelGGL((mpd_unfold_kernel<h16_t, float>), g, b, 0, st, p);
  else if (grad_dtype == EVT_DT_F32 && dsrc_dtype == EVT_DT_HALF) hipLaunchKernelGGL((mpd_unfold_kernel<float, h16_t>), g, b, 0, st, p);
  else return EVT_EINVAL;
  return evt_check_launch();
}

}  // extern "C"
