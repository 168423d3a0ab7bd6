// Multi-tensor weight preparation (weight-norm fold + GEMM-ready layouts), fused losses, AdamW over a
// flat arena and small element-wise fusions of the s2 path.  gfx950 only.
//
// Reference call sites (file:line under /root/reference):
//   weight_norm            src/easevoice/module/modules.py:162,174,184,228-296; models.py:427-436,486-536,563-574
//   stage mean             src/easevoice/module/models.py:457-466
//   gated activation       src/easevoice/module/commons.py:94-101
//   feature/LSGAN losses   src/easevoice/module/losses.py:7-43
//   AdamW, grad norm       src/train/sovits.py:294-319,505,522; src/easevoice/module/commons.py:140-155
#include "evt_common.h"
#include "../../include/evt.h"

namespace {

__device__ __forceinline__ void store_w(void* base, long idx, float w, int dtype) {
  if (dtype == EVT_DT_HALF) reinterpret_cast<h16_t*>(base)[idx] = f2h(w);
  else reinterpret_cast<float*>(base)[idx] = w;
}

__device__ __forceinline__ long reg_idx(const evt_wlayout& L, int d0, int d1, int kk) {
  const int chunk = d1 / L.reg_ck, cc = d1 - chunk * L.reg_ck;
  return (((long)d0 * L.reg_nchunk + chunk) * L.reg_kp + kk) * L.reg_ck + cc;
}

__device__ __forceinline__ long alt_idx(const evt_wlayout& L, int d0, int d1, int kk) {
  const int chunk = d0 / L.alt_ck, cc = d0 - chunk * L.alt_ck;
  if (L.stride == 1) {
    const int t = L.k - 1 - kk;
    return (((long)d1 * L.alt_nchunk + chunk) * L.alt_kp + t) * L.alt_ck + cc;
  }
  const int J = (L.k + L.stride - 1) / L.stride;
  const int ph = kk % L.stride, j = kk / L.stride;
  const int jp = J - 1 - j;
  return ((((long)ph * L.d1 + d1) * L.alt_nchunk + chunk) * L.alt_kp + jp) * L.alt_ck + cc;
}

// Workgroup -> table row: consecutive blockIdx values round-robin over the 8 XCDs, so a plain row = blockIdx mapping
// would spread the 32 rows that complete one 64-byte ALT run (cc = d0 % 32) over 8 different L2s.  Rows are handed out
// to an XCD in runs of 32 instead, so those partial-line stores meet in one L2 and leave it as whole lines.
__device__ __forceinline__ int fold_row_of_block(int b, int nrows) {
  const int xcd = b & 7, slot = b >> 3;
  const int row = ((slot >> 5) * 8 + xcd) * 32 + (slot & 31);
  return row < nrows ? row : -1;
}

// Round 5, tried and dropped (profiles/r05_fold_tiled_negative.txt): one block per 32 consecutive rows of a layer, every wave
// transposing its own 64-column tiles through a wave-private LDS square so that the ALT image leaves as whole 64-byte runs
// (no block barrier after the norms).  Same images (tested), 3.3 x SLOWER: 1126 us against 337 us per launch, the s2 step
// 23.25 -> 24.9 ms.  The row-per-block form wins on parallelism -- 12 K blocks of 256 threads with ~20 elements each, against
// 375 blocks whose waves walk 20 tiles of 96 memory instructions in sequence -- and both forms issue the same number of
// 2-byte store instructions per element; a version that pays off has to make the stores 16 bytes per lane for BOTH images
// (REG runs are 8 consecutive input channels at a stride of k floats in the source row), which needs tiles of 8 k columns.
__global__ __launch_bounds__(256) void wn_fold_kernel(const evt_wprep_item* items, const int32_t* rows, int nrows) {
  __shared__ float red[4];
  const int trow = fold_row_of_block(blockIdx.x, nrows);
  if (trow < 0) return;
  const evt_wprep_item it = items[rows[2 * trow]];
  const int d0 = rows[2 * trow + 1];
  const evt_wlayout& L = it.lay;
  const int n = (it.src_d1 ? it.src_d1 : L.d1) * L.k;     // src_d1: the image has more (zero) columns than the parameter
  const float* v = it.v + (long)d0 * n;
  float scale = 1.f;
  if (it.g) {
    float ss = 0.f;
    for (int e = threadIdx.x; e < n; e += 256) ss += v[e] * v[e];
    ss = block_reduce_sum_256(ss, red);
    scale = it.g[d0] / sqrtf(ss);
  }
  for (int e = threadIdx.x; e < n; e += 256) {
    const int d1 = e / L.k, kk = e - d1 * L.k;
    const float w = v[e] * scale;
    if (it.reg) store_w(it.reg, reg_idx(L, d0, d1, kk), w, it.dtype);
    if (it.alt) store_w(it.alt, alt_idx(L, d0, d1, kk), w, it.dtype);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// wn_fold8_kernel (round 6): the same fold, EIGHT consecutive output rows per block, both images leaving in 16-byte pieces.
//
// What bounds wn_fold_kernel (0.34 ms per model and launch at 1.16 TB/s, 0.145 of the HBM rate): the ALT image is indexed
// [d1][chunk(d0)][tap][d0 % ck], so the elements of ONE source row land 2 bytes at a time in n different 64-byte runs --
// every wave store touches 64 lines, and a run is completed by 32 different blocks; the REG image [d0][chunk(d1)][tap]
// [d1 % ck] is a permutation of the row, stored as 2-byte pieces 64 bytes apart.  The pass is bound by the number of L2
// write requests, not by bytes.  With the rows d0 .. d0 + 7 of one layer in a block (same ALT chunk, consecutive d0 % ck):
//   * the eight values of a (d1, tap) pair are ONE 16-byte ALT piece  -> 8 x fewer ALT requests;
//   * REG: the row's elements of a segment of d1 columns are permuted in LDS and leave as contiguous 16-byte pieces.
// The round-5 attempt at this (32 rows per block, wave-private transposes, 375 fat blocks) was 3.3 x slower: too few blocks.
// Here a block is 8 rows (1.5 K blocks per model), a segment is 64 columns (a few KB of LDS: many blocks per CU).
// Bit-identical images: the norm of a row is accumulated in wn_fold_kernel's order (thread-strided partial sums,
// block_reduce_sum_256), the scaled value is the same product, the rounding the same f2h.
// Groups the fast path does not take (fewer than 8 rows, fp32 images, rows not 16-byte aligned, an image missing) run the
// per-row body on each of their rows.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int F8_ROW = 1536;            // staged source elements per row and segment: 8 rows x 1536 x 2 B = 24 KiB per block (48 KiB: 152 us, 24 KiB: 142 us)
constexpr int F8_KMAX = 16;             // taps (incl. padding) of the fast path

__device__ __forceinline__ void wn_fold_row(const evt_wprep_item& it, int d0, float* red) {
  const evt_wlayout& L = it.lay;
  const int n = (it.src_d1 ? it.src_d1 : L.d1) * L.k;
  const float* v = it.v + (long)d0 * n;
  float scale = 1.f;
  if (it.g) {
    float ss = 0.f;
    for (int e = threadIdx.x; e < n; e += 256) ss += v[e] * v[e];
    ss = block_reduce_sum_256(ss, red);
    scale = it.g[d0] / sqrtf(ss);
  }
  for (int e = threadIdx.x; e < n; e += 256) {
    const int d1 = e / L.k, kk = e - d1 * L.k;
    const float w = v[e] * scale;
    if (it.reg) store_w(it.reg, reg_idx(L, d0, d1, kk), w, it.dtype);
    if (it.alt) store_w(it.alt, alt_idx(L, d0, d1, kk), w, it.dtype);
  }
}

// block -> group: consecutive blockIdx values go round-robin over the 8 XCDs; the four groups that complete a 64-byte ALT
// run (32 rows) are handed to ONE XCD so that the run leaves its L2 as a whole line
__device__ __forceinline__ int fold_group_of_block(int b, int ngroups) {
  const int xcd = b & 7, slot = b >> 3;
  const int grp = ((slot >> 2) * 8 + xcd) * 4 + (slot & 3);
  return grp < ngroups ? grp : -1;
}

__global__ __launch_bounds__(256) void wn_fold8_kernel(const evt_wprep_item* items, const int32_t* groups, int ngroups) {
  __shared__ float red[4];
  __shared__ float red8[8][4];
  __shared__ float scale_s[8];
  // the segment's scaled 16-bit values in SOURCE order, [row][column * k + tap]: written 8 bytes per lane, both images
  // are gathered from it (2-byte LDS reads are cheap; 2-byte LDS WRITES 64 bytes apart were a 32-way bank conflict)
  __shared__ __attribute__((aligned(16))) h16_t src_s[8][F8_ROW];
  const int grp = fold_group_of_block(blockIdx.x, ngroups);
  if (grp < 0) return;
  const evt_wprep_item it = items[groups[3 * grp]];
  const int d0 = groups[3 * grp + 1], nr = groups[3 * grp + 2];
  const evt_wlayout& L = it.lay;
  const int d1n = it.src_d1 ? it.src_d1 : L.d1;
  const int n = d1n * L.k;
  const bool fast = nr == 8 && it.dtype == EVT_DT_HALF && it.reg && it.alt && (d0 & 7) == 0 && (L.alt_ck & 7) == 0 &&
                    (n & 3) == 0 && L.reg_kp <= F8_KMAX && L.k <= F8_KMAX && L.reg_ck * L.k <= F8_ROW &&
                    (L.reg_ck & 7) == 0 && (L.d1 % L.reg_ck) == 0 && ((L.reg_ck * L.k) & 3) == 0;
  if (!fast) {
    for (int r = 0; r < nr; ++r) {
      wn_fold_row(it, d0 + r, red);
      __syncthreads();
    }
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // ---- norms of the eight rows, each in wn_fold_kernel's summation order (thread-strided partial sums, wave tree, then
  //      the four wave sums left to right), the eight rows' loads interleaved ----
  if (it.g) {
    float ss[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) ss[r] = 0.f;
    const float* v0 = it.v + (long)d0 * n;
    for (int e = tid; e < n; e += 256) {
#pragma unroll
      for (int r = 0; r < 8; ++r) { const float x = v0[(long)r * n + e]; ss[r] += x * x; }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float w = wave_reduce_sum(ss[r]);
      if (lane == 0) red8[r][wv] = w;
    }
    __syncthreads();
    if (tid < 8) scale_s[tid] = it.g[d0 + tid] / sqrtf(red8[tid][0] + red8[tid][1] + red8[tid][2] + red8[tid][3]);
  } else if (tid < 8) {
    scale_s[tid] = 1.f;
  }
  __syncthreads();
  const int k = L.k, kp = L.reg_kp, ck = L.reg_ck;
  h16_t* regp = reinterpret_cast<h16_t*>(it.reg);
  h16_t* altp = reinterpret_cast<h16_t*>(it.alt);
  const long reg_row = (long)L.reg_nchunk * kp * ck;                       // elements per REG row
  // columns per segment: as many whole REG chunks as the staging holds (k = 5: 288 columns, k = 11: 128, k = 1: 1536)
  const int seg_max = (F8_ROW / (ck * k)) * ck;
  for (int c0 = 0; c0 < L.d1; c0 += seg_max) {                             // (L.d1 >= d1n: padded columns stay zero)
    const int seg = min(seg_max, L.d1 - c0);                               // a multiple of reg_ck
    const int src_cols = max(0, min(seg, d1n - c0));                       // columns of this segment the parameter has
    const int segk = src_cols * k;                                         // source elements per row (a multiple of 4)
    // ---- source -> LDS: 4 consecutive elements of one row per lane, scaled and rounded ----
    const int f4_per_row = segk >> 2;
    for (int q = tid; q < 8 * f4_per_row; q += 256) {
      const int r = q / f4_per_row, f = q - r * f4_per_row;
      const f32x4 x = *reinterpret_cast<const f32x4*>(it.v + (long)(d0 + r) * n + (long)c0 * k + 4 * f);
      const float sc = scale_s[r];
      h16_t w[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) w[u] = f2h(x[u] * sc);
      *reinterpret_cast<uint2*>(&src_s[r][4 * f]) = *reinterpret_cast<uint2*>(w);
    }
    __syncthreads();
    // ---- REG: row r's segment is the contiguous range [chunk0 * kp * ck, + seg * kp) of its image row; a 16-byte piece
    //      = 8 consecutive columns of one (chunk, tap).  Padding taps / columns are written as zeros. ----
    {
      const int pieces = seg * kp / 8;
      const int ck8 = ck >> 3;
      for (int q = tid; q < 8 * pieces; q += 256) {
        const int r = q / pieces, pc = q - r * pieces;
        const int cc8 = pc % ck8, t2 = pc / ck8;
        const int kk = t2 % kp, chl = t2 / kp;
        const int dl0 = chl * ck + cc8 * 8;                                 // first of the piece's 8 columns
        h16_t w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = (kk < k && dl0 + j < src_cols) ? src_s[r][(dl0 + j) * k + kk] : (h16_t)0;
        *reinterpret_cast<uint4*>(regp + (long)(d0 + r) * reg_row + (long)(c0 / ck) * kp * ck + pc * 8) =
            *reinterpret_cast<uint4*>(w);
      }
    }
    // ---- ALT: one 16-byte piece per (column, tap): rows d0 .. d0 + 7 are consecutive d0 % alt_ck ----
    for (int q = tid; q < segk; q += 256) {
      const int dl = q / k, kk = q - dl * k;
      h16_t w[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) w[j] = src_s[j][q];
      *reinterpret_cast<uint4*>(altp + alt_idx(L, d0, c0 + dl, kk)) = *reinterpret_cast<uint4*>(w);
    }
    __syncthreads();
  }
}

// A deterministic split-K weight gradient (evt_conv1d_bwd_weight_parts) leaves its partial sums in slabs: slab 0 is the
// image itself, slabs 1 .. n-1 are `stride` floats apart in `extra`.  This folds one d0-row of them into slab 0, in
// index order, in place (the row belongs to this block alone).  Image order, 16 bytes per lane, eight slabs requested
// before the first is added: the pass is a stream of n x row bytes (2.4 GB per s2 step), not a chain of dependent loads.
__device__ __forceinline__ void fold_slabs_row(float* dw_row, const float* extra_row, long stride, int n, int row_floats) {
  const int nv = row_floats >> 2;                  // rows are multiples of 32 floats (ck = 32) or handled by the tail loop
  for (int v = threadIdx.x; v < nv; v += 256) {
    f32x4 s = reinterpret_cast<const f32x4*>(dw_row)[v];
    int k = 1;
    for (; k + 8 <= n; k += 8) {
      f32x4 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = reinterpret_cast<const f32x4*>(extra_row + (long)(k - 1 + u) * stride)[v];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += t[u];
    }
    for (; k < n; ++k) s += reinterpret_cast<const f32x4*>(extra_row + (long)(k - 1) * stride)[v];
    reinterpret_cast<f32x4*>(dw_row)[v] = s;
  }
  for (int e = (nv << 2) + threadIdx.x; e < row_floats; e += 256) {
    float s = dw_row[e];
    for (int k = 1; k < n; ++k) s += extra_row[(long)(k - 1) * stride + e];
    dw_row[e] = s;
  }
  __syncthreads();                                 // the row is read back below by other threads of the block
}

__global__ __launch_bounds__(256) void wn_grad_kernel(const evt_wprep_item* items, const int32_t* rows) {
  __shared__ float red[4];
  const evt_wprep_item it = items[rows[2 * blockIdx.x]];
  if (!it.dw) return;     // no gradient image: e.g. a member of a packed projection, whose .grad the pack's rows update
  const int d0 = rows[2 * blockIdx.x + 1];
  const evt_wlayout& L = it.lay;
  const int n = (it.src_d1 ? it.src_d1 : L.d1) * L.k;
  const float* v = it.v + (long)d0 * n;
  float* dv = it.dv + (long)d0 * n;
  if (it.used) {
    const int nu = it.used[0], nb = it.used[1];
    if (nu > 1 && it.dw_extra) {
      const int row_floats = L.reg_nchunk * L.reg_kp * L.reg_ck;
      const long ro = (long)d0 * row_floats;
      fold_slabs_row(const_cast<float*>(it.dw) + ro, it.dw_extra + ro, it.dw_part_stride, nu, row_floats);
    }
    if (nb > 0 && it.db_part && it.db && threadIdx.x == 0) {
      float s = 0.f;
      for (int k = 0; k < nb; ++k) s += it.db_part[(long)k * L.d0 + d0];
      it.db[d0] += s;
    }
  }
  if (!it.g) {
    for (int e = threadIdx.x; e < n; e += 256) {
      const int d1 = e / L.k, kk = e - d1 * L.k;
      dv[e] += it.dw[reg_idx(L, d0, d1, kk)];
    }
    return;
  }
  float ss = 0.f, dot = 0.f;
  for (int e = threadIdx.x; e < n; e += 256) {
    const int d1 = e / L.k, kk = e - d1 * L.k;
    const float x = v[e];
    ss += x * x;
    dot += x * it.dw[reg_idx(L, d0, d1, kk)];
  }
  ss = block_reduce_sum_256(ss, red);
  dot = block_reduce_sum_256(dot, red);
  const float norm = sqrtf(ss);
  const float gval = it.g[d0];
  // w = g * v / |v|  =>  dg = <dw, v>/|v| ;  dv = g/|v| * (dw - v <dw, v>/|v|^2)
  if (threadIdx.x == 0) it.dg[d0] += dot / norm;
  const float s1 = gval / norm, s2 = dot / ss;
  for (int e = threadIdx.x; e < n; e += 256) {
    const int d1 = e / L.k, kk = e - d1 * L.k;
    dv[e] += s1 * (it.dw[reg_idx(L, d0, d1, kk)] - v[e] * s2);
  }
}

template <typename T>
__global__ void add3_scale_kernel(const T* a, const T* b, const T* c, float scale, T* out, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = to_f<T>(a[i]);
    if (b) v += to_f<T>(b[i]);
    if (c) v += to_f<T>(c[i]);
    out[i] = from_f<T>(v * scale);
  }
}

// out = dy * act'(y) with the derivative expressed through the activation OUTPUT y; 8 (bf16) / 4 (fp32) elements per lane
template <typename T>
__global__ void dact_mul_kernel(const T* dy, const T* y, int kind, float slope, T* out, long n) {
  constexpr int V = 16 / sizeof(T);
  const long nv = n / V;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
    uint4 d = reinterpret_cast<const uint4*>(dy)[i];
    const uint4 a = reinterpret_cast<const uint4*>(y)[i];
    T* dh = reinterpret_cast<T*>(&d);
    const T* ah = reinterpret_cast<const T*>(&a);
#pragma unroll
    for (int e = 0; e < V; ++e) dh[e] = from_f<T>(to_f<T>(dh[e]) * dact_from_out(kind, to_f<T>(ah[e]), slope));
    reinterpret_cast<uint4*>(out)[i] = d;
  }
  if (blockIdx.x == 0 && threadIdx.x < n - nv * V) {
    const long i = nv * V + threadIdx.x;
    out[i] = from_f<T>(to_f<T>(dy[i]) * dact_from_out(kind, to_f<T>(y[i]), slope));
  }
}

// out = leaky_relu(x): the activated copy the HiFi-GAN residual units feed to their convolutions (16 bytes per lane)
template <typename T>
__global__ void lrelu_kernel(const T* x, float slope, T* out, long n) {
  constexpr int V = 16 / sizeof(T);
  const long nv = n / V;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
    uint4 v = reinterpret_cast<const uint4*>(x)[i];
    T* h = reinterpret_cast<T*>(&v);
#pragma unroll
    for (int e = 0; e < V; ++e) h[e] = from_f<T>(lrelu_f(to_f<T>(h[e]), slope));
    reinterpret_cast<uint4*>(out)[i] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x < n - nv * V) {
    const long i = nv * V + threadIdx.x;
    out[i] = from_f<T>(lrelu_f(to_f<T>(x[i]), slope));
  }
}

__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + __expf(-x)); }

template <typename T>
__global__ void gated_fwd_kernel(const T* xin, const T* g, T* acts, int nseq, int len, int H) {
  const long total = (long)nseq * len * H;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int h = (int)(i % H);
    const long nt = i / H;
    const int s = (int)(nt / len);
    float a = to_f<T>(xin[nt * 2 * H + h]), b = to_f<T>(xin[nt * 2 * H + H + h]);
    if (g) { a += to_f<T>(g[(long)s * 2 * H + h]); b += to_f<T>(g[(long)s * 2 * H + H + h]); }
    acts[i] = from_f<T>(tanhf(a) * sigmoid_f(b));
  }
}

template <typename T>
__global__ void gated_bwd_kernel(const T* xin, const T* g, const T* dacts, T* dxin, float* dg, int nseq, int len,
                                 int H) {
  const long total = (long)nseq * len * H;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int h = (int)(i % H);
    const long nt = i / H;
    const int s = (int)(nt / len);
    float a = to_f<T>(xin[nt * 2 * H + h]), b = to_f<T>(xin[nt * 2 * H + H + h]);
    if (g) { a += to_f<T>(g[(long)s * 2 * H + h]); b += to_f<T>(g[(long)s * 2 * H + H + h]); }
    const float t = tanhf(a), sg = sigmoid_f(b), d = to_f<T>(dacts[i]);
    const float da = d * sg * (1.f - t * t), db = d * t * sg * (1.f - sg);
    dxin[nt * 2 * H + h] = from_f<T>(da);
    dxin[nt * 2 * H + H + h] = from_f<T>(db);
    if (dg) { atomicAdd(dg + (long)s * 2 * H + h, da); atomicAdd(dg + (long)s * 2 * H + H + h, db); }
  }
}

// the same backward when the conditioning gradient is wanted: dg[s][c] = sum over the sequence's positions.  A block takes
// `tch` positions of ONE sequence, a thread keeps a channel pair and adds its positions in registers: one atomic per
// (block, channel) instead of one per element (1.2 M atomics on 6144 addresses per WN layer: 12.5 us for 6 MB of traffic)
template <typename T>
__global__ __launch_bounds__(256) void gated_bwd_dg_kernel(const T* xin, const T* g, const T* dacts, T* dxin, float* dg,
                                                           int nseq, int len, int H, int tch) {
  const int per_seq = (len + tch - 1) / tch;
  const int s = blockIdx.x / per_seq;
  const int t0 = (blockIdx.x - s * per_seq) * tch, t1 = min(len, t0 + tch);
  for (int h = threadIdx.x; h < H; h += 256) {
    const float ga = to_f<T>(g[(long)s * 2 * H + h]), gb = to_f<T>(g[(long)s * 2 * H + H + h]);
    float sa = 0.f, sb = 0.f;
    for (int t = t0; t < t1; ++t) {
      const long nt = (long)s * len + t;
      const float a = to_f<T>(xin[nt * 2 * H + h]) + ga, b = to_f<T>(xin[nt * 2 * H + H + h]) + gb;
      const float th = tanhf(a), sg = sigmoid_f(b), d = to_f<T>(dacts[nt * H + h]);
      const float da = d * sg * (1.f - th * th), db = d * th * sg * (1.f - sg);
      dxin[nt * 2 * H + h] = from_f<T>(da);
      dxin[nt * 2 * H + H + h] = from_f<T>(db);
      sa += da;
      sb += db;
    }
    atomicAdd(dg + (long)s * 2 * H + h, sa);
    atomicAdd(dg + (long)s * 2 * H + H + h, sb);
  }
}

// ---- fused loss reductions over a table of segments ------------------------------------------------
// The segments (37 feature maps from 10 K to 21 M elements) are treated as ONE flat index space split evenly over the
// blocks: a block walks the part of each segment that falls into its range with 16-byte loads and finishes with a
// single atomic (the first version launched 64 blocks per segment and paid ~2400 same-address atomics, ~46 ns each).
constexpr int SEG_MAX = 64;

template <typename T, int MODE>
__device__ __forceinline__ float seg_term(float a, float b, float target) {
  if (MODE == 0) return fabsf(a - b);
  const float d = target - a;
  return d * d;
}
template <typename T, int MODE>
__device__ __forceinline__ float seg_grad(float a, float b, float target, float dl) {
  if (MODE == 0) { const float d = a - b; return d > 0.f ? dl : (d < 0.f ? -dl : 0.f); }
  return -2.f * (target - a) * dl;
}

template <typename T, int MODE, bool BWD>
__global__ __launch_bounds__(256) void seg_flat_kernel(const evt_seg* segs, int nseg, float target, float* out,
                                                       const float* dloss) {
  constexpr int V = 16 / sizeof(T);
  __shared__ float red[4];
  __shared__ long pre[SEG_MAX + 1];
  if (threadIdx.x == 0) {
    long acc = 0;
    for (int i = 0; i < nseg; ++i) { pre[i] = acc; acc += segs[i].n; }
    pre[nseg] = acc;
  }
  __syncthreads();
  const long total = pre[nseg];
  long chunk = (total + gridDim.x - 1) / gridDim.x;
  chunk = (chunk + 2047) / 2048 * 2048;
  const long lo = (long)blockIdx.x * chunk, hi = min(total, lo + chunk);
  float acc = 0.f;
  for (int si = 0; si < nseg; ++si) {
    const long s0 = pre[si], s1 = pre[si + 1];
    if (s1 <= lo || s0 >= hi) continue;
    const evt_seg s = segs[si];
    if (BWD && !s.da) continue;
    const T* a = reinterpret_cast<const T*>(s.a);
    const T* b = reinterpret_cast<const T*>(s.b);
    T* da = reinterpret_cast<T*>(s.da);
    const long i0 = max(lo, s0) - s0, i1 = min(hi, s1) - s0;      // element range inside this segment
    const float dl = BWD ? dloss[0] * s.scale : 0.f;
    float part = 0.f;
    // head up to the next multiple of V, vector body, tail
    const long v0 = min(i1, (i0 + V - 1) / V * V), v1 = v0 + (i1 - v0) / V * V;
    const bool vec_ok = ((((uintptr_t)a) | (MODE == 0 ? (uintptr_t)b : 0) | (BWD ? (uintptr_t)da : 0)) & 15) == 0;
    if (vec_ok) {
      for (long i = i0 + threadIdx.x; i < v0; i += 256) {
        const float av = to_f<T>(a[i]), bv = MODE == 0 ? to_f<T>(b[i]) : 0.f;
        if (BWD) da[i] = from_f<T>(seg_grad<T, MODE>(av, bv, target, dl)); else part += seg_term<T, MODE>(av, bv, target);
      }
      for (long i = v0 + (long)threadIdx.x * V; i < v1; i += 256L * V) {
        const uint4 va = *reinterpret_cast<const uint4*>(a + i);
        uint4 vb = make_uint4(0, 0, 0, 0);
        if (MODE == 0) vb = *reinterpret_cast<const uint4*>(b + i);
        const T* pa = reinterpret_cast<const T*>(&va);
        const T* pb = reinterpret_cast<const T*>(&vb);
        uint4 vo;
        T* po = reinterpret_cast<T*>(&vo);
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const float av = to_f<T>(pa[e]), bv = MODE == 0 ? to_f<T>(pb[e]) : 0.f;
          if (BWD) po[e] = from_f<T>(seg_grad<T, MODE>(av, bv, target, dl)); else part += seg_term<T, MODE>(av, bv, target);
        }
        if (BWD) *reinterpret_cast<uint4*>(da + i) = vo;
      }
      for (long i = v1 + threadIdx.x; i < i1; i += 256) {
        const float av = to_f<T>(a[i]), bv = MODE == 0 ? to_f<T>(b[i]) : 0.f;
        if (BWD) da[i] = from_f<T>(seg_grad<T, MODE>(av, bv, target, dl)); else part += seg_term<T, MODE>(av, bv, target);
      }
    } else {
      for (long i = i0 + threadIdx.x; i < i1; i += 256) {
        const float av = to_f<T>(a[i]), bv = MODE == 0 ? to_f<T>(b[i]) : 0.f;
        if (BWD) da[i] = from_f<T>(seg_grad<T, MODE>(av, bv, target, dl)); else part += seg_term<T, MODE>(av, bv, target);
      }
    }
    acc += part * s.scale;
  }
  if (!BWD) {
    acc = block_reduce_sum_256(acc, red);
    if (threadIdx.x == 0 && acc != 0.f) atomicAdd(out, acc);
  }
}

// ---- AdamW over a flat arena ---------------------------------------------------------------------------
__global__ void adamw_flat_kernel(float* p, const float* g, float* m, float* v, const evt_adamw_seg* segs, int nseg,
                                  float b1, float b2, float eps, float bc1, float bc2_sqrt, float gscale,
                                  long lo, long hi) {
  for (long i = lo + blockIdx.x * (long)blockDim.x + threadIdx.x; i < hi; i += (long)gridDim.x * blockDim.x) {
    int si = -1;
    for (int s = 0; s < nseg; ++s)
      if (i >= segs[s].begin && i < segs[s].end) { si = s; break; }
    if (si < 0) continue;
    const float lr = segs[si].lr, wd = segs[si].weight_decay;
    const float gr = g[i] * gscale;
    float pv = p[i];
    pv *= 1.f - lr * wd;
    const float mv = b1 * m[i] + (1.f - b1) * gr;
    const float vv = b2 * v[i] + (1.f - b2) * gr * gr;
    m[i] = mv; v[i] = vv;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    pv -= (lr / bc1) * (mv / denom);
    p[i] = pv;
  }
}

__global__ void counter_inc_kernel(int* c, const float* skip) {
  if (skip && *skip != 0.f) return;
  *c += 1;
}

// same update, bias corrections from a step number held in device memory (graph-replayable: no per-step host argument)
__global__ void adamw_flat_dev_kernel(float* p, const float* g, float* m, float* v, const evt_adamw_seg* segs, int nseg,
                                      float b1, float b2, float eps, const int* stepp, float gscale, long lo, long hi,
                                      const float* skip) {
  if (skip && *skip != 0.f) return;      // GradScaler.step on an overflow: the optimiser is not stepped at all
  const float stepf = (float)*stepp;
  const float bc1 = 1.f - powf(b1, stepf);
  const float bc2_sqrt = sqrtf(1.f - powf(b2, stepf));
  for (long i = lo + blockIdx.x * (long)blockDim.x + threadIdx.x; i < hi; i += (long)gridDim.x * blockDim.x) {
    int si = -1;
    for (int s = 0; s < nseg; ++s)
      if (i >= segs[s].begin && i < segs[s].end) { si = s; break; }
    if (si < 0) continue;
    const float lr = segs[si].lr, wd = segs[si].weight_decay;
    const float gr = g[i] * gscale;
    float pv = p[i];
    pv *= 1.f - lr * wd;
    const float mv = b1 * m[i] + (1.f - b1) * gr;
    const float vv = b2 * v[i] + (1.f - b2) * gr * gr;
    m[i] = mv; v[i] = vv;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    pv -= (lr / bc1) * (mv / denom);
    p[i] = pv;
  }
}

// GradScaler.unscale_: g *= 1 / scale, any non-finite element raises the flag (the flag is only ever written with 1)
__global__ __launch_bounds__(256) void unscale_check_kernel(float* g, long n, const float* scale, float* found) {
  const float inv = (float)(1.0 / (double)*scale);
  bool bad = false;
  const long n4 = n >> 2;
  float4* g4 = reinterpret_cast<float4*>(g);
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 v = g4[i];
    bad |= !(isfinite(v.x) && isfinite(v.y) && isfinite(v.z) && isfinite(v.w));
    v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
    g4[i] = v;
  }
  for (long i = (n4 << 2) + blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float v = g[i];
    bad |= !isfinite(v);
    g[i] = v * inv;
  }
  if (bad) *found = 1.f;
}

struct ScalerFlags { float* f[4]; };
__global__ void scaler_update_kernel(float* scale, int* tracker, ScalerFlags fl, int nflags, float growth, float backoff,
                                     int interval) {
  bool found = false;
  for (int i = 0; i < nflags; ++i) {
    found |= *fl.f[i] != 0.f;
    *fl.f[i] = 0.f;
  }
  if (found) {
    *scale = *scale * backoff;
    *tracker = 0;
  } else {
    const int ok = *tracker + 1;
    if (ok == interval) {
      const float ns = *scale * growth;
      if (isfinite(ns)) *scale = ns;
      *tracker = 0;
    } else {
      *tracker = ok;
    }
  }
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* x, long n, float* out) {
  __shared__ float red[4];
  float acc = 0.f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) acc += x[i] * x[i];
  acc = block_reduce_sum_256(acc, red);
  if (threadIdx.x == 0) atomicAdd(out, acc);
}

inline int grid_for(long n, int cap = 8192) {
  long b = (n + 255) / 256;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

#include <stdarg.h>
#include <stdio.h>
// profiling aid (evt.h: evt_debug_kernel_tags): nothing is recorded unless a profiler switched it on
static bool g_tags_on = false;
static thread_local char g_last_tag[128] = "";
extern "C" void evt_set_last_tag(const char* fmt, ...) {
  if (!g_tags_on) return;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_tag, sizeof(g_last_tag), fmt, ap);
  va_end(ap);
}

extern "C" {

const char* evt_last_kernel_tag(void) { return g_last_tag; }
void evt_debug_kernel_tags(int32_t enable) { g_tags_on = enable != 0; }

#ifndef EVT_SRC_HASH
#define EVT_SRC_HASH "unknown"
#endif
const char* evt_version(void) { return "evt-hip 0.3 (gfx950, half = " EVT_HALF_NAME ") src=" EVT_SRC_HASH; }
int32_t evt_half_dtype(void) { return EVT_DT_HALF; }

int evt_wn_fold_multi(const evt_wprep_item* items, const int32_t* row_index, int32_t nrows, void* stream) {
  if (!items || !row_index || nrows <= 0) return EVT_EINVAL;
  const int nblocks = ((nrows + 255) / 256) * 256;        // whole 8 x 32 row groups (fold_row_of_block)
  hipLaunchKernelGGL(wn_fold_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, items, row_index, nrows);
  return evt_check_launch();
}

int evt_wn_fold_groups(const evt_wprep_item* items, const int32_t* group_index, int32_t ngroups, void* stream) {
  if (!items || !group_index || ngroups <= 0) return EVT_EINVAL;
  const int nblocks = ((ngroups + 31) / 32) * 32;         // whole 8 x 4 group runs (fold_group_of_block)
  hipLaunchKernelGGL(wn_fold8_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, items, group_index, ngroups);
  return evt_check_launch();
}

int evt_wn_grad_multi(const evt_wprep_item* items, const int32_t* row_index, int32_t nrows, void* stream) {
  if (!items || !row_index || nrows <= 0) return EVT_EINVAL;
  hipLaunchKernelGGL(wn_grad_kernel, dim3(nrows), dim3(256), 0, (hipStream_t)stream, items, row_index);
  return evt_check_launch();
}

int evt_add3_scale(int32_t dtype, const void* a, const void* b, const void* c, float scale, void* out, int64_t n,
                   void* stream) {
  if (!a || !out || n <= 0) return EVT_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EVT_DT_HALF)
    hipLaunchKernelGGL(add3_scale_kernel<h16_t>, dim3(grid_for(n)), dim3(256), 0, st, (const h16_t*)a,
                       (const h16_t*)b, (const h16_t*)c, scale, (h16_t*)out, (long)n);
  else if (dtype == EVT_DT_F32)
    hipLaunchKernelGGL(add3_scale_kernel<float>, dim3(grid_for(n)), dim3(256), 0, st, (const float*)a, (const float*)b,
                       (const float*)c, scale, (float*)out, (long)n);
  else return EVT_EINVAL;
  return evt_check_launch();
}

int evt_leaky_relu(int32_t dtype, const void* x, float slope, void* out, int64_t n, void* stream) {
  if (!x || !out || n <= 0) return EVT_EINVAL;
  if (((uintptr_t)x | (uintptr_t)out) & 15) return EVT_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EVT_DT_HALF)
    hipLaunchKernelGGL(lrelu_kernel<h16_t>, dim3(grid_for((n + 7) / 8)), dim3(256), 0, st, (const h16_t*)x, slope,
                       (h16_t*)out, (long)n);
  else if (dtype == EVT_DT_F32)
    hipLaunchKernelGGL(lrelu_kernel<float>, dim3(grid_for((n + 3) / 4)), dim3(256), 0, st, (const float*)x, slope,
                       (float*)out, (long)n);
  else return EVT_EINVAL;
  return evt_check_launch();
}

int evt_dact_mul(int32_t dtype, const void* dy, const void* y, int32_t act_kind, float slope, void* out, int64_t n,
                 void* stream) {
  if (!dy || !y || !out || n <= 0) return EVT_EINVAL;
  if (((uintptr_t)dy | (uintptr_t)y | (uintptr_t)out) & 15) return EVT_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EVT_DT_HALF)
    hipLaunchKernelGGL(dact_mul_kernel<h16_t>, dim3(grid_for((n + 7) / 8)), dim3(256), 0, st, (const h16_t*)dy,
                       (const h16_t*)y, act_kind, slope, (h16_t*)out, (long)n);
  else if (dtype == EVT_DT_F32)
    hipLaunchKernelGGL(dact_mul_kernel<float>, dim3(grid_for((n + 3) / 4)), dim3(256), 0, st, (const float*)dy,
                       (const float*)y, act_kind, slope, (float*)out, (long)n);
  else return EVT_EINVAL;
  return evt_check_launch();
}

int evt_gated_act_fwd(int32_t dtype, const void* xin, const void* g, void* acts, int32_t nseq, int32_t len, int32_t H,
                      void* stream) {
  if (!xin || !acts || nseq <= 0 || len <= 0 || H <= 0) return EVT_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const long n = (long)nseq * len * H;
  if (dtype == EVT_DT_HALF)
    hipLaunchKernelGGL(gated_fwd_kernel<h16_t>, dim3(grid_for(n)), dim3(256), 0, st, (const h16_t*)xin,
                       (const h16_t*)g, (h16_t*)acts, nseq, len, H);
  else if (dtype == EVT_DT_F32)
    hipLaunchKernelGGL(gated_fwd_kernel<float>, dim3(grid_for(n)), dim3(256), 0, st, (const float*)xin,
                       (const float*)g, (float*)acts, nseq, len, H);
  else return EVT_EINVAL;
  return evt_check_launch();
}

int evt_gated_act_bwd(int32_t dtype, const void* xin, const void* g, const void* dacts, void* dxin, float* dg,
                      int32_t nseq, int32_t len, int32_t H, void* stream) {
  if (!xin || !dacts || !dxin || nseq <= 0 || len <= 0 || H <= 0) return EVT_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const long n = (long)nseq * len * H;
  if (dg && g) {
    // positions per block: >= 512 blocks when the problem has them, at least 4 positions each
    int tch = (int)(((long)nseq * len + 511) / 512);
    if (tch < 4) tch = 4;
    const int blocks = nseq * ((len + tch - 1) / tch);
    if (dtype == EVT_DT_HALF)
      hipLaunchKernelGGL(gated_bwd_dg_kernel<h16_t>, dim3(blocks), dim3(256), 0, st, (const h16_t*)xin, (const h16_t*)g,
                         (const h16_t*)dacts, (h16_t*)dxin, dg, nseq, len, H, tch);
    else if (dtype == EVT_DT_F32)
      hipLaunchKernelGGL(gated_bwd_dg_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)xin, (const float*)g,
                         (const float*)dacts, (float*)dxin, dg, nseq, len, H, tch);
    else return EVT_EINVAL;
    return evt_check_launch();
  }
  if (dtype == EVT_DT_HALF)
    hipLaunchKernelGGL(gated_bwd_kernel<h16_t>, dim3(grid_for(n)), dim3(256), 0, st, (const h16_t*)xin,
                       (const h16_t*)g, (const h16_t*)dacts, (h16_t*)dxin, dg, nseq, len, H);
  else if (dtype == EVT_DT_F32)
    hipLaunchKernelGGL(gated_bwd_kernel<float>, dim3(grid_for(n)), dim3(256), 0, st, (const float*)xin,
                       (const float*)g, (const float*)dacts, (float*)dxin, dg, nseq, len, H);
  else return EVT_EINVAL;
  return evt_check_launch();
}

#define SEG_LAUNCH(MODE, BWD, TARGET, OUT, DLOSS)                                                                   \
  do {                                                                                                             \
    if (nseg > SEG_MAX) return EVT_ENOTSUP;                                                                        \
    if (dtype == EVT_DT_HALF)                                                                                      \
      hipLaunchKernelGGL((seg_flat_kernel<h16_t, MODE, BWD>), dim3(512), dim3(256), 0, st, segs, nseg, TARGET, OUT, DLOSS); \
    else if (dtype == EVT_DT_F32)                                                                                  \
      hipLaunchKernelGGL((seg_flat_kernel<float, MODE, BWD>), dim3(512), dim3(256), 0, st, segs, nseg, TARGET, OUT, DLOSS);  \
    else return EVT_EINVAL;                                                                                        \
  } while (0)

int evt_l1_multi_fwd(int32_t dtype, const evt_seg* segs, int32_t nseg, float* out, void* stream) {
  if (!segs || nseg <= 0 || !out) return EVT_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  SEG_LAUNCH(0, false, 0.f, out, (const float*)nullptr);
  return evt_check_launch();
}
int evt_l1_multi_bwd(int32_t dtype, const evt_seg* segs, int32_t nseg, const float* dloss, void* stream) {
  if (!segs || nseg <= 0 || !dloss) return EVT_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  SEG_LAUNCH(0, true, 0.f, (float*)nullptr, dloss);
  return evt_check_launch();
}
int evt_lsgan_multi_fwd(int32_t dtype, const evt_seg* segs, int32_t nseg, float target, float* out, void* stream) {
  if (!segs || nseg <= 0 || !out) return EVT_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  SEG_LAUNCH(1, false, target, out, (const float*)nullptr);
  return evt_check_launch();
}
int evt_lsgan_multi_bwd(int32_t dtype, const evt_seg* segs, int32_t nseg, float target, const float* dloss,
                        void* stream) {
  if (!segs || nseg <= 0 || !dloss) return EVT_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  SEG_LAUNCH(1, true, target, (float*)nullptr, dloss);
  return evt_check_launch();
}

int evt_adamw_flat(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                   const evt_adamw_seg* segs, int32_t nseg, float beta1, float beta2, float eps, int32_t step,
                   float grad_scale, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !segs || nseg <= 0 || nseg > 64 || step <= 0 || n <= 0)
    return EVT_EINVAL;
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adamw_flat_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                     exp_avg_sq, segs, nseg, beta1, beta2, eps, bc1, bc2_sqrt, grad_scale, 0L, (long)n);
  return evt_check_launch();
}

int evt_adamw_flat_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                       const evt_adamw_seg* segs, int32_t nseg, float beta1, float beta2, float eps,
                       int32_t* step_counter, float grad_scale, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !segs || nseg <= 0 || nseg > 64 || !step_counter || n <= 0)
    return EVT_EINVAL;
  hipLaunchKernelGGL(counter_inc_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_counter, (const float*)nullptr);
  hipLaunchKernelGGL(adamw_flat_dev_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                     exp_avg_sq, segs, nseg, beta1, beta2, eps, (const int*)step_counter, grad_scale, 0L, (long)n,
                     (const float*)nullptr);
  return evt_check_launch();
}

int evt_adamw_flat_dev_guarded(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                               const evt_adamw_seg* segs, int32_t nseg, float beta1, float beta2, float eps,
                               int32_t* step_counter, float grad_scale, const float* skip, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !segs || nseg <= 0 || nseg > 64 || !step_counter || n <= 0)
    return EVT_EINVAL;
  hipLaunchKernelGGL(counter_inc_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_counter, skip);
  hipLaunchKernelGGL(adamw_flat_dev_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                     exp_avg_sq, segs, nseg, beta1, beta2, eps, (const int*)step_counter, grad_scale, 0L, (long)n, skip);
  return evt_check_launch();
}

int evt_scaler_unscale(float* grad, int64_t n, const float* scale, float* found_inf, void* stream) {
  if (!grad || !scale || !found_inf || n <= 0 || ((uintptr_t)grad & 15)) return EVT_EINVAL;
  hipLaunchKernelGGL(unscale_check_kernel, dim3(grid_for(n >> 2, 4096)), dim3(256), 0, (hipStream_t)stream, grad, (long)n,
                     scale, found_inf);
  return evt_check_launch();
}

int evt_scaler_update(float* scale, int32_t* growth_tracker, float* const* found_inf, int32_t nflags, float growth_factor,
                      float backoff_factor, int32_t growth_interval, void* stream) {
  if (!scale || !growth_tracker || !found_inf || nflags <= 0 || nflags > 4 || growth_interval <= 0) return EVT_EINVAL;
  ScalerFlags fl{};
  for (int i = 0; i < nflags; ++i) {
    if (!found_inf[i]) return EVT_EINVAL;
    fl.f[i] = found_inf[i];
  }
  hipLaunchKernelGGL(scaler_update_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, scale, growth_tracker, fl, nflags,
                     growth_factor, backoff_factor, growth_interval);
  return evt_check_launch();
}

int evt_adamw_flat_dev_range(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t lo, int64_t hi,
                             const evt_adamw_seg* segs, int32_t nseg, float beta1, float beta2, float eps,
                             int32_t* step_counter, int32_t bump, float grad_scale, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !segs || nseg <= 0 || nseg > 64 || !step_counter || lo < 0 || hi <= lo)
    return EVT_EINVAL;
  if (bump)
    hipLaunchKernelGGL(counter_inc_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_counter, (const float*)nullptr);
  hipLaunchKernelGGL(adamw_flat_dev_kernel, dim3(grid_for(hi - lo)), dim3(256), 0, (hipStream_t)stream, param, grad,
                     exp_avg, exp_avg_sq, segs, nseg, beta1, beta2, eps, (const int*)step_counter, grad_scale, (long)lo,
                     (long)hi, (const float*)nullptr);
  return evt_check_launch();
}

int evt_sumsq(const float* x, int64_t n, float* out, void* stream) {
  if (!x || !out || n <= 0) return EVT_EINVAL;
  hipLaunchKernelGGL(sumsq_kernel, dim3(grid_for(n, 2048)), dim3(256), 0, (hipStream_t)stream, x, (long)n, out);
  return evt_check_launch();
}

}  // extern "C"
