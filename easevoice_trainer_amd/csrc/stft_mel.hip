// STFT magnitude -> mel -> log of the generated waveform, forward and analytic backward (gfx950).
//
// Replaces torch.stft + matmul + log at src/easevoice/module/mel_processing.py:93-142 (called on y_hat
// at src/train/sovits.py:480-489, so it needs a backward): reflect-pad (n_fft-hop)/2, hann window,
// n_fft = 2048 real DFT per frame, sqrt(re^2+im^2+1e-6), [n_mels x 1025] mel matmul, log(clamp(.,1e-5)).
//
// One workgroup (256 threads, 4 waves) per frame.  The 2048-point radix-2 DIT FFT keeps 8 complex values
// per thread (element e = c*256 + tid): butterflies of span 1..32 exchange partners with wavefront
// shuffles (ds-free), span 64/128 go through LDS once each, span 256..1024 are register-local.
// Twiddles come from a per-block LDS table built with sincospif (no fast-math trig on the parity path).
#include "evt_common.h"
#include "../../include/evt.h"

namespace {

constexpr int NFFT = 2048;
constexpr int NBIN = NFFT / 2 + 1;

struct cplx { float x, y; };
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }

// in: v[c] = a[bitrev(c*256+t)]; out: v[c] = FFT(a)[c*256+t]   (forward sign e^{-i...})
__device__ __forceinline__ void fft2048(cplx v[8], const cplx* tw, cplx* ex, int t) {
  // spans 1..32: partner is lane t^h of the same wave
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    const int h = 1 << s;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int e = c * 256 + t;
      cplx p;
      p.x = __shfl_xor(v[c].x, h, 64);
      p.y = __shfl_xor(v[c].y, h, 64);
      const cplx w = tw[(e & (h - 1)) << (10 - s)];
      if ((e & h) == 0) { const cplx m = cmul(w, p); v[c] = {v[c].x + m.x, v[c].y + m.y}; }
      else { const cplx m = cmul(w, v[c]); v[c] = {p.x - m.x, p.y - m.y}; }
    }
  }
  // spans 64, 128: partner lives in another wave -> one LDS exchange per stage
#pragma unroll
  for (int s = 6; s < 8; ++s) {
    const int h = 1 << s;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 8; ++c) ex[c * 256 + t] = v[c];
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int e = c * 256 + t;
      const cplx p = ex[e ^ h];
      const cplx w = tw[(e & (h - 1)) << (10 - s)];
      if ((e & h) == 0) { const cplx m = cmul(w, p); v[c] = {v[c].x + m.x, v[c].y + m.y}; }
      else { const cplx m = cmul(w, v[c]); v[c] = {p.x - m.x, p.y - m.y}; }
    }
  }
  // spans 256, 512, 1024: both halves of the butterfly are in this thread's registers
#pragma unroll
  for (int s = 8; s < 11; ++s) {
    const int h = 1 << s, hc = h >> 8;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if ((c & hc) == 0) {
        const int e = c * 256 + t;
        const cplx w = tw[(e & (h - 1)) << (10 - s)];
        const cplx m = cmul(w, v[c + hc]);
        v[c + hc] = {v[c].x - m.x, v[c].y - m.y};
        v[c] = {v[c].x + m.x, v[c].y + m.y};
      }
    }
  }
}

__device__ __forceinline__ void build_twiddles(cplx* tw, int t) {
  for (int k = t; k < NFFT / 2; k += 256) {
    float s, c;
    sincospif(-2.0f * (float)k / (float)NFFT, &s, &c);  // exp(-2 pi i k / N)
    tw[k] = {c, s};
  }
}

__device__ __forceinline__ int reflect_index(int pi, int pad, int wav_len) {
  int j = pi - pad;
  if (j < 0) j = -j;
  else if (j >= wav_len) j = 2 * (wav_len - 1) - j;
  return j;
}

// workspace per frame: re[NBIN], im[NBIN], melpre[n_mels]
__global__ __launch_bounds__(256) void mel_fwd_kernel(const float* wav, const float* window, const float* basis,
                                                      float* spec_out, float* mel_out, float* ws, int wav_len,
                                                      int frames, int hop, int n_mels) {
  __shared__ cplx tw[NFFT / 2];
  __shared__ cplx ex[NFFT];
  __shared__ float mag[NBIN + 3];
  const int t = threadIdx.x;
  const int f = blockIdx.x % frames, seq = blockIdx.x / frames;
  const int pad = (NFFT - hop) / 2;
  build_twiddles(tw, t);
  float* xs = reinterpret_cast<float*>(ex);
  for (int n = t; n < NFFT; n += 256)
    xs[n] = wav[(long)seq * wav_len + reflect_index(f * hop + n, pad, wav_len)] * window[n];
  __syncthreads();
  cplx v[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) v[c] = {xs[__brev((unsigned)(c * 256 + t)) >> 21], 0.f};
  fft2048(v, tw, ex, t);
  float* wsf = ws + (long)blockIdx.x * (2 * NBIN + n_mels);
#pragma unroll
  for (int c = 0; c < 5; ++c) {
    const int e = c * 256 + t;
    if (e < NBIN) {
      const float m = sqrtf(v[c].x * v[c].x + v[c].y * v[c].y + 1e-6f);
      mag[e] = m;
      wsf[e] = v[c].x;
      wsf[NBIN + e] = v[c].y;
      if (spec_out) spec_out[((long)seq * NBIN + e) * frames + f] = m;
    }
  }
  __syncthreads();
  // mel: two threads per mel bin (even / odd k), combined with one shuffle
  for (int m0 = 0; m0 < n_mels; m0 += 128) {
    const int m = m0 + (t >> 1);
    float acc = 0.f;
    if (m < n_mels) {
      const float* br = basis + (long)m * NBIN;
      for (int k = (t & 1); k < NBIN; k += 2) acc += br[k] * mag[k];
    }
    acc += __shfl_xor(acc, 1, 64);
    if (m < n_mels && (t & 1) == 0) {
      wsf[2 * NBIN + m] = acc;
      mel_out[((long)seq * n_mels + m) * frames + f] = logf(fmaxf(acc, 1e-5f));
    }
  }
}

__global__ __launch_bounds__(256) void mel_bwd_kernel(const float* dmel, const float* window, const float* basis,
                                                      const float* ws, float* dwav, int wav_len, int frames, int hop,
                                                      int n_mels) {
  __shared__ cplx tw[NFFT / 2];
  __shared__ cplx ex[NFFT];
  __shared__ float dpre[512];
  const int t = threadIdx.x;
  const int f = blockIdx.x % frames, seq = blockIdx.x / frames;
  const int pad = (NFFT - hop) / 2;
  build_twiddles(tw, t);
  const float* wsf = ws + (long)blockIdx.x * (2 * NBIN + n_mels);
  // d(log(clamp(p,1e-5)))/dp = 1/p for p > 1e-5 else 0
  for (int m = t; m < n_mels; m += 256) {
    const float p = wsf[2 * NBIN + m];
    dpre[m] = p > 1e-5f ? dmel[((long)seq * n_mels + m) * frames + f] / p : 0.f;
  }
  __syncthreads();
  // G_k = dmag_k * (re, im)/mag for k <= N/2, 0 above; stored conjugated for the forward-sign FFT
  for (int k = t; k < NFFT; k += 256) {
    cplx g = {0.f, 0.f};
    if (k < NBIN) {
      float dm = 0.f;
      for (int m = 0; m < n_mels; ++m) dm += basis[(long)m * NBIN + k] * dpre[m];
      const float re = wsf[k], im = wsf[NBIN + k];
      const float inv = dm / sqrtf(re * re + im * im + 1e-6f);
      g = {re * inv, -(im * inv)};
    }
    ex[k] = g;
  }
  __syncthreads();
  cplx v[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) v[c] = ex[__brev((unsigned)(c * 256 + t)) >> 21];
  fft2048(v, tw, ex, t);
  // dx_n = Re(FFT(conj G)_n); window, then scatter through the reflect padding
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int n = c * 256 + t;
    const float d = v[c].x * window[n];
    atomicAdd(dwav + (long)seq * wav_len + reflect_index(f * hop + n, pad, wav_len), d);
  }
}

}  // namespace

extern "C" {

int64_t evt_mel_workspace_floats(int32_t nseq, int32_t wav_len, int32_t n_fft, int32_t hop, int32_t n_mels) {
  const int pad = (n_fft - hop) / 2;
  const int frames = (wav_len + 2 * pad - n_fft) / hop + 1;
  return (int64_t)nseq * frames * (2 * (n_fft / 2 + 1) + n_mels);
}

int evt_mel_fwd(const float* wav, const float* window, const float* mel_basis, float* spec_out, float* mel_out,
                float* ws, int32_t nseq, int32_t wav_len, int32_t n_fft, int32_t hop, int32_t n_mels, void* stream) {
  if (!wav || !window || !mel_basis || !mel_out || !ws || nseq <= 0) return EVT_EINVAL;
  if (n_fft != NFFT || n_mels > 512 || n_mels <= 0 || hop <= 0 || hop > n_fft) return EVT_ENOTSUP;
  const int pad = (n_fft - hop) / 2;
  if (wav_len <= pad) return EVT_EINVAL;  // reflect padding needs pad < wav_len
  const int frames = (wav_len + 2 * pad - n_fft) / hop + 1;
  if (frames <= 0) return EVT_EINVAL;
  hipLaunchKernelGGL(mel_fwd_kernel, dim3(nseq * frames), dim3(256), 0, (hipStream_t)stream, wav, window, mel_basis,
                     spec_out, mel_out, ws, wav_len, frames, hop, n_mels);
  return evt_check_launch();
}

int evt_mel_bwd(const float* dmel, const float* window, const float* mel_basis, const float* ws, float* dwav,
                int32_t nseq, int32_t wav_len, int32_t n_fft, int32_t hop, int32_t n_mels, void* stream) {
  if (!dmel || !window || !mel_basis || !ws || !dwav || nseq <= 0) return EVT_EINVAL;
  if (n_fft != NFFT || n_mels > 512 || n_mels <= 0 || hop <= 0 || hop > n_fft) return EVT_ENOTSUP;
  const int pad = (n_fft - hop) / 2;
  if (wav_len <= pad) return EVT_EINVAL;
  const int frames = (wav_len + 2 * pad - n_fft) / hop + 1;
  if (frames <= 0) return EVT_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(dwav, 0, (size_t)nseq * wav_len * sizeof(float), st) != hipSuccess) return EVT_ELAUNCH;
  hipLaunchKernelGGL(mel_bwd_kernel, dim3(nseq * frames), dim3(256), 0, st, dmel, window, mel_basis, ws, dwav, wav_len,
                     frames, hop, n_mels);
  return evt_check_launch();
}

}  // extern "C"
