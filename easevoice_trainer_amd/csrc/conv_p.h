// Launch descriptor of the implicit-GEMM conv kernels (conv1d.hip: conv_igemm, conv_deep.hip: conv_deep).
#pragma once
#include "evt_common.h"
#include "../../include/evt.h"
#include <cstdlib>

namespace evt_conv {

struct ConvP {
  const void* x;     // K-side operand, [nseq][Lin][Cin]
  const void* xact;  // optional activation OUTPUT with x's shape: x_eff = x * dact(xact)
  const void* w;     // prepared weights [phase][Cout][nchunk][KHp][CK]
  const float* bias; // [Cout] or null
  const void* res;   // [nseq][Lout][Cout] or null (added last)
  const void* gate;  // [nseq][Lout][Cout] or null: result *= (gate > 0 ? 1 : gate_slope) before res
  void* y;           // [nseq][Lout][Cout]
  int nseq, Lin, Lout, Cin, Cout;
  int KHp;           // taps in the prepared image (zero padded)
  int s_in, dil, off_in;
  int s_out, off_out, off_out_phase;
  int Q, U;          // q per sequence, units per sequence
  int nchunk;
  long w_phase_stride;  // elements
  float in_slope;
  int xact_kind;
  float xact_slope;
  int out_act;
  float out_slope;
  float gate_slope;
  int P, Y;          // position blocks, channel tiles
};

// Weight-gradient descriptor: dW[a][chunk(b)][tap][cc] += sum_{seq,q} A[seq][q][a] * B[seq][q*s + tap*dil + off][b]
struct WgP {
  const void* A;      // [nseq][LA][CA]   q-indexed operand
  const void* Aact;   // optional activation output (A_eff = A * dact(Aact))
  const void* B;      // [nseq][LB][CB]   tap-shifted operand
  const void* Bact;
  float* dw;          // [CA][nchunk][KHp][CK] fp32
  int nseq, LA, LB, CA, CB;
  int KH, KHp, s, dil, off, Q;
  int nchunk;
  float a_slope, b_slope;     // lrelu-on-load slopes (1 = identity)
  int aact_kind, bact_kind;
  float aact_slope, bact_slope;
  int nsplit;
  int ntapgrp;
  int xcd_order;      // wgrad_gemm: blocks decoded so that one split's tiles share an XCD (set by its launcher)
  float* dbias;       // optional: += column sums of A_eff (only valid when A is dy)
  // deterministic split-K (evt_conv1d_bwd_weight_parts, wgrad_epi.h); parts == 0: classic fp32 atomics into dw
  float* dw_extra;    // slabs 1 .. parts-1 of the gradient image, part_stride floats apart (slab 0 is dw)
  long part_stride;
  int parts;          // slabs the caller provides = upper bound for nsplit
  int prev_used;      // slabs already holding partial sums of this step (0: first launch); slabs >= prev_used (and the
                      // matching rows of db_part) are stored, the others added to
  int now_used;       // max(prev_used, nsplit): what the kernel writes to used[]
  int dirty0;         // slab 0 already holds sums of this step (any earlier launch): add to it instead of storing
  float* db_part;     // [parts][CA] partial bias gradients, or null (dbias then takes atomics)
  int* used;          // device int32[2]: {slabs of dw in use, slabs of db_part in use}, written by the kernel
  int* used_host;     // HOST int the launcher sets to now_used (0 stays for kernels without slab support)
  // scratch for the kernels that finish their reduction with a second launch (fold.hip); null: fp32 atomics
  float* ws;
  long ws_floats;
};

// conv_deep.hip: GEMM-grade path for wide bf16 layers (K-side channels % 64 == 0, output channels % 128 == 0, no
// load-side fusion).  Returns EVT_ENOTSUP when the descriptor does not qualify.
bool deep_eligible(const ConvP& p, int dtype, int out_ch, int k_ch, int nphase);
int launch_conv_deep(const ConvP& p, int out_ch, int k_ch, int nphase, hipStream_t st);
// conv_narrow.hip: weights-in-registers kernel for the stride-1 C = 16 / 32 vocoder layers (plain operands)
bool narrow_eligible(const ConvP& p, int dtype, int out_ch, int k_ch, int nphase);
int launch_conv_narrow(const ConvP& p, int out_ch, int k_ch, int nphase, hipStream_t st);
// weight gradient on the same LDS-DMA structure (A channels % 128 == 0, B channels % 32 == 0, plain operands, no dbias)
bool wgrad_deep_eligible(const WgP& p, int dtype);
int launch_wgrad_deep(const WgP& p, hipStream_t st);
// dense-layer (k = 1) weight gradient as a 128 x 128 GEMM tile (A, B channels % 128 == 0, long reductions); fuses dbias
bool wgrad_gemm_eligible(const WgP& p, int dtype);
int launch_wgrad_gemm(const WgP& p, hipStream_t st);
// ring-pipelined variant for the latency-bound mid-size layers (A channels % 64 == 0); fuses dbias
bool wgrad_ring_eligible(const WgP& p, int dtype);
int launch_wgrad_ring(const WgP& p, hipStream_t st);
// wgrad_halo.hip: stride-1 layers with 3..11 taps whose sequences are long enough for sequence-local K stages: all taps
// of a block read ONE staged window of the shifted operand (vocoder stages, WN layers, encoder FFN); fuses dbias
bool wgrad_halo_eligible(const WgP& p, int dtype);
int launch_wgrad_halo(const WgP& p, hipStream_t st);
// fold.hip: out[e] += sum over b < nb of part[b * stride + e], e < n, rows added in a fixed order
int launch_fold_partials(const float* part, long stride, int nb, float* out, long n, hipStream_t st);
// how the launchers split the positions: nsplit and stages per split from the stage count, the tile count and p.parts
// (classic mode: `target` blocks; slab mode: additionally nsplit <= parts)
void wgrad_pick_split(const WgP& p, long tiles, int nstages, long target, int min_stages, int* nsplit, int* per);

// rows_gemm.hip: k = 1 layers applied to at most 16 rows in total (one vector per batch item)
bool rows16_eligible(const struct evt_conv1d_params* c, int rows, int n_out, int k_red, bool fused);
int launch_rows16(const void* x, const void* w, const float* bias, void* y, int M, int N, int K, hipStream_t st);

}  // namespace evt_conv
