/*
 * evt.h — C ABI of libevt_hip.so, the MI355X (gfx950) compute library behind the
 * GPT-SoVITS training hot path of megaease/easevoice-trainer.
 *
 * The reference has NO native code: every entry point below replaces a vendor kernel that
 * the reference reaches through torch (ATen / cuDNN / cuBLAS / cuFFT) at the cited call
 * site.  Citations are file:line under /root/reference.
 *
 * Conventions
 *   - plain C, raw device pointers, sizes as int32/int64, `stream` is a hipStream_t passed
 *     as void*; no torch types, no hidden allocation, no global mutable state; every call
 *     only enqueues work on `stream` and returns 0 or a positive errno-style code.
 *   - activations are CHANNELS-LAST: a "sequence tensor" is [nseq][len][channels] with the
 *     channel index contiguous (the reference's [B, C, L] transposed); 16-bit tensors are raw
 *     uint16 storage.  dtype: 0 = float32, 1 = bfloat16, 2 = float16 (fp32 accumulation everywhere).
 *   - the same sources are built twice: libevt_hip.so serves dtype 0 and 1, libevt_hip_f16.so serves
 *     dtype 0 and 2 (the reference's `fp16_run` autocast type, src/train/sovits.py:459-525); both export
 *     exactly this header; a half code the loaded build does not serve returns ENOTSUP (95).
 *     evt_half_dtype() says which one a loaded library is.
 *   - weight gradients / bias gradients are fp32 and are ACCUMULATED (+=) into the caller's
 *     buffers (atomics), so the caller zeroes them once per optimiser step.
 */
#ifndef EVT_H
#define EVT_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: exactly the declarations of this header are its exported symbols */
#pragma GCC visibility push(default)

#define EVT_DT_F32 0
#define EVT_DT_BF16 1
#define EVT_DT_F16 2
#define EVT_ACT_NONE 0
#define EVT_ACT_LRELU 1
#define EVT_ACT_TANH 2
#define EVT_IMPL_AUTO 0
#define EVT_IMPL_NAIVE 1 /* direct-form reference kernels (any shape, groups) */
#define EVT_IMPL_IGEMM 2 /* LDS-tiled implicit-GEMM on MFMA */

/* "evt-hip <version> (gfx950) src=<12 hex digits>": the digits are a hash of the sources the library was built from
 * (easevoice_trainer_amd/build.py::source_hash); the Python binding refuses a library whose hash differs from the
 * sources next to it. */
const char* evt_version(void);
/* the 16-bit dtype code this build serves: EVT_DT_BF16 (libevt_hip.so) or EVT_DT_F16 (libevt_hip_f16.so) */
int32_t evt_half_dtype(void);
/* Profiling aid, OUTSIDE the contract above and off by default: after evt_debug_kernel_tags(1) every dispatcher
 * records the name of the kernel instantiation it launched in a per-thread buffer that evt_last_kernel_tag() returns
 * (bench.py's roofline leg groups its timings by these names, the same names rocprofv3 prints).  With tags off (the
 * product path) nothing is recorded.  evt_debug_* are the library's ONLY process-global mutable state (two flags and a
 * thread-local name buffer): measurement switches, never read by a computation's arithmetic.  DEBUG ONLY, NOT REENTRANT:
 * switching them while another thread is inside the library is undefined; the product path never calls them. */
void evt_debug_kernel_tags(int32_t enable);
const char* evt_last_kernel_tag(void);
/* measurement switch of the bf16 attention kernels: 1 = both query/key tiles of a wave in one instruction stream
 * , 0 = one tile at a time (fewer registers, more waves per SIMD; default).  Same results either way. */
void evt_debug_attn_variant(int32_t joint);

/* Scratch memory the caller must provide, in bytes, for the ops that take a workspace pointer; -1 for an unknown op or
 * a wrong dims count.  The library never allocates.
 *   EVT_WS_MEL          dims = {nseq, wav_len, n_fft, hop, n_mels}   ws of evt_mel_fwd / evt_mel_bwd
 *   EVT_WS_ATTN_BWD     dims = {B, H, L}                             delta_ws of evt_attn_prefixlm_bwd
 *   EVT_WS_MHA_BWD      dims = {B, H, Tq}                            delta_ws of evt_mha_bwd
 *   EVT_WS_MASKED_KL    dims = {}                                    out2 of evt_masked_kl_fwd (sum, live frames)
 *   EVT_WS_NONE         every other entry point: 0 */
#define EVT_WS_NONE 0
#define EVT_WS_MEL 1
#define EVT_WS_ATTN_BWD 2
#define EVT_WS_MHA_BWD 3
#define EVT_WS_MASKED_KL 4
int64_t evt_workspace_bytes(int32_t op, const int64_t* dims, int32_t ndims);

/* ---------------------------------------------------------------------------------------
 * Conv1d / ConvTranspose1d family.
 * Replaces F.conv1d / F.conv_transpose1d / F.conv2d((k,1)) reached from
 *   HiFi-GAN Generator + ResBlock1   src/easevoice/module/models.py:452-471, modules.py:298-311
 *   WN (enc_q, flow)                 src/easevoice/module/modules.py:187-212
 *   FFN convs                        src/easevoice/module/attentions.py:408-416
 *   DiscriminatorS / DiscriminatorP  src/easevoice/module/models.py:538-587
 * One fused op computes, per output position:
 *     y = act_out( conv( lrelu(x, in_slope) ) + bias ) * 1 + res
 * DiscriminatorP's Conv2d (k,1)/(s,1) over [B,1,T/p,p] is the same op on B*p sequences
 * (the caller lays the period axis out as the sequence axis).
 * ------------------------------------------------------------------------------------- */
typedef struct evt_conv1d_params {
  int32_t dtype;      /* EVT_DT_*: activations and prepared weights */
  int32_t nseq;       /* sequences (batch, or batch*period) */
  int32_t lin;        /* input length per sequence */
  int32_t cin, cout;  /* module in/out channels (ConvTranspose: in = weight dim 0) */
  int32_t k, stride, pad, dil, groups;
  int32_t transposed; /* 0 Conv1d (weight [cout][cin/g][k]); 1 ConvTranspose1d (weight [cin][cout][k]) */
  float in_slope;     /* leaky-relu slope applied to x on load; 1.0f = identity */
  int32_t out_act;    /* EVT_ACT_* applied to conv+bias */
  float out_slope;    /* slope when out_act == EVT_ACT_LRELU */
  int32_t impl;       /* EVT_IMPL_* */
} evt_conv1d_params;

/* output length for these hyper-parameters (torch semantics) */
/* 1 when the backward of this conv runs on the LDS-DMA GEMM path (conv_deep), which takes dy with the output
 * activation's derivative already applied (evt_dact_mul) and out_act = EVT_ACT_NONE in the backward calls; else 0. */
int32_t evt_conv1d_wants_plain_dy(const evt_conv1d_params* c);

/* 1 when the FORWARD of this conv (in_slope != 1: leaky-relu on load, the HiFi-GAN upsamplers of models.py:457-460) would
 * run on the LDS-DMA GEMM path if it were handed the activated input: the caller then applies evt_leaky_relu once,
 * calls forward and weight gradient with in_slope = 1 on the activated tensor, and backward-data with in_slope and the
 * activated tensor as `x` (leaky-relu keeps the sign, the derivative is the same).  Else 0. */
int32_t evt_conv1d_wants_plain_x(const evt_conv1d_params* c);

int32_t evt_conv1d_lout(const evt_conv1d_params* p);

/* Prepared-weight layouts.  The parameter tensor is W[d0][d1][k] (Conv1d: d0=cout, d1=cin/g;
 * ConvTranspose1d: d0=cin, d1=cout).  Two GEMM-ready images are derived from it:
 *   REG: [d0][chunk(d1)][k'][ck]                — "rows = d0"; Conv1d forward, ConvT backward-data, all dW
 *   ALT: stride==1: [d1][chunk(d0)][k' flipped][ck]; stride>1: polyphase [phase][d1][chunk(d0)][j][ck]
 *                                                — Conv1d backward-data, ConvT forward
 * evt_conv1d_layout fills the geometry; element counts are in elements of the compute dtype
 * (REG/ALT) or fp32 (dW, same geometry as REG). */
typedef struct evt_wlayout {
  int32_t d0, d1, k, stride;
  int32_t reg_ck, reg_nchunk, reg_kp;          /* REG image */
  int32_t alt_ck, alt_nchunk, alt_kp, alt_nphase; /* ALT image (alt_kp = taps per phase) */
  int64_t reg_elems, alt_elems;
} evt_wlayout;
int evt_conv1d_layout(const evt_conv1d_params* p, evt_wlayout* out);

/* Multi-tensor weight preparation: one launch for a whole model.
 * Replaces torch.nn.utils.weight_norm's per-forward w = g * v / ||v|| (dims 1,2 per d0 row;
 * modules.py:162,174,184,228-296, models.py:427-436,486-536,563-574) plus the cast/layout
 * change into REG and ALT images.  `g` may be NULL for plain (non weight-normed) convs. */
typedef struct evt_wprep_item {
  const float* v;   /* [d0][d1][k] fp32 master (weight_v, or weight when g == NULL) */
  const float* g;   /* [d0] fp32 weight_g or NULL */
  void* reg;        /* REG image (compute dtype) or NULL */
  void* alt;        /* ALT image (compute dtype) or NULL */
  /* backward side (evt_wn_grad_multi): */
  const float* dw;  /* fp32 REG-geometry gradient, accumulated by evt_conv1d_bwd_weight */
  float* dv;        /* [d0][d1][k] fp32, += */
  float* dg;        /* [d0] fp32, += (NULL when g == NULL) */
  evt_wlayout lay;
  int32_t dtype;
  int32_t src_d1;   /* 0, or the parameter's own d1 when lay.d1 was padded up (enc_q.pre: 1025 spectrogram bins in an image
                     * of 1088 columns); v / dv rows then hold src_d1 * k values, the image's extra columns stay zero */
  /* slabs of a deterministic split-K weight gradient (evt_conv1d_bwd_weight_parts); all NULL / 0 without them */
  const float* dw_extra;   /* slabs 1.. of the gradient image, dw_part_stride floats apart (slab 0 is dw) */
  int64_t dw_part_stride;
  const float* db_part;    /* [slab][d0] partial bias gradients or NULL */
  float* db;               /* [d0] fp32 bias gradient, += sum of the db_part slabs in use */
  const int32_t* used;     /* DEVICE int32[2] = {slabs of dw in use (0 or 1: dw only), slabs of db_part in use}: written
                            * by the weight-gradient kernels, reset to 0 by the caller together with dw */
} evt_wprep_item;
/* items is a DEVICE pointer to n items; row_index is a DEVICE int32 [nrows][2] table of
 * (item, d0-row) pairs, one workgroup each. */
int evt_wn_fold_multi(const evt_wprep_item* items, const int32_t* row_index, int32_t nrows, void* stream);
/* The same fold over GROUPS of up to eight consecutive d0-rows of one item: group_index is a DEVICE int32 [ngroups][3] table
 * of (item, first d0-row, rows in the group <= 8), one workgroup each.  A full group of a 16-bit image pair leaves as
 * 16-byte pieces (csrc/elementwise.hip: wn_fold8_kernel); the images are bit-identical to evt_wn_fold_multi's. */
int evt_wn_fold_groups(const evt_wprep_item* items, const int32_t* group_index, int32_t ngroups, void* stream);
int evt_wn_grad_multi(const evt_wprep_item* items, const int32_t* row_index, int32_t nrows, void* stream);

/* y[nseq][lout][cout] = act_out(conv(lrelu(x)) + bias) + res ; bias/res may be NULL.
 * The MFMA path reads REG for Conv1d and ALT for ConvTranspose1d; the direct path reads REG.
 * Pass both images (either may be NULL if the path that needs it cannot be selected). */
int evt_conv1d_fwd(const evt_conv1d_params* p, const void* x, const void* w_reg, const void* w_alt,
                   const float* bias, const void* res, void* y, void* stream);

/* dx[nseq][lin][cin] = convT( dy * act_out'(y) ) * lrelu'(x) + dx_add
 * y may be NULL when out_act == NONE; x may be NULL when in_slope == 1; dx_add may be NULL.
 * The MFMA path reads ALT for Conv1d and REG for ConvTranspose1d; the direct path reads REG. */
int evt_conv1d_bwd_data(const evt_conv1d_params* p, const void* dy, const void* y, const void* w_reg,
                        const void* w_alt, const void* x, const void* dx_add, void* dx, void* stream);

/* dW(REG geometry, fp32) += ..., dbias[cout] += sum(dy * act_out'(y)); dbias may be NULL. */
int evt_conv1d_bwd_weight(const evt_conv1d_params* p, const void* x, const void* dy, const void* y,
                          float* dw, float* dbias, void* stream);

/* The same gradient with a DETERMINISTIC split over the positions.  The MFMA weight-gradient kernels split the
 * reduction over several blocks per output tile; evt_conv1d_bwd_weight combines their partial tiles with fp32 atomics
 * (sum order = finishing order of the blocks: last bits differ from run to run, and every partial tile crosses the
 * fabric as atomic packets).  Here split s owns SLAB s of the gradient image -- slab 0 is `dw` itself, slabs
 * 1 .. parts-1 are dw_extra + (s-1) * part_stride -- and writes it with plain stores: added to slabs that already hold sums of this
 * step (slab 0 when dirty0, slabs below prev_used), stored to the others.  The fused bias gradient goes to db_part[s][cout] the same way (when db_part is NULL: atomics into
 * dbias as before).  evt_wn_grad_multi adds the slabs in index order.  The kernel records the slab counts in
 * used_dev[0..1]; `used` returns the count on the host for the next launch's prev_used.  used == 0 on return: the
 * kernel that handled this shape has no slab mode and accumulated into dw / dbias the classic way.
 * torch.autograd accumulates weight gradients in one fixed order per op; this restores that property. */
typedef struct evt_wgrad_parts {
  float* dw_extra;
  int64_t part_stride;   /* floats */
  float* db_part;
  int32_t* used_dev;
  int32_t parts;         /* slabs provided (>= 1) = upper bound of the split */
  int32_t prev_used;     /* in: slabs already holding partial sums since the last reset (0: none) */
  int32_t used;          /* out */
  int32_t dirty0;        /* in: 1 when slab 0 (dw) may already hold sums of this step -- an earlier launch of any kind since
                          * the caller zeroed it -- and has to be added to; 0: slab 0 is stored like the others (no read) */
  float* ws;             /* scratch (or NULL) for the kernels whose result is tiny -- the Cout = 1 / Cin = 1 layers, the 16-
                          * and 32-channel vocoder stages: their blocks store partial results here and a second launch adds
                          * them into dw / dbias in a fixed order, instead of one fp32 atomic per block and address.  Any
                          * size >= 1 MiB helps (the kernels bound their block count by it); contents need not survive
                          * the call; one buffer per stream. */
  int64_t ws_floats;
} evt_wgrad_parts;
int evt_conv1d_bwd_weight_parts(const evt_conv1d_params* p, const void* x, const void* dy, const void* y,
                                float* dw, float* dbias, evt_wgrad_parts* sp, void* stream);

/* One HiFi-GAN ResBlock1 step  y = x + c2(lrelu(c1(lrelu(x), dilation d)))  (modules.py:299-308 of the reference; both
 * convolutions C -> C, kernel k, "same" padding, c2 undilated) as ONE launch for the narrow vocoder stages: bf16,
 * C in {16, 32}, k in {3, 7, 11}, d in 1..5.  x, y, xa, mid_a: [nseq][L][C]; w1_reg / w2_reg: the REG images of the two
 * convolutions; b1 / b2 fp32 [C] or NULL.  xa = lrelu(x) and mid_a = lrelu(c1(xa) + b1) are the operands the backward
 * launches (evt_conv1d_bwd_*) take; pass NULL to skip writing them (inference). */
typedef struct evt_resunit_params {
  int32_t dtype, nseq, L, C, k, dil;
  float slope;
} evt_resunit_params;
int32_t evt_resunit_supported(const evt_resunit_params* p);
int evt_resunit_fwd(const evt_resunit_params* p, const void* x, const void* w1_reg, const void* w2_reg, const float* b1,
                    const float* b2, void* xa, void* mid_a, void* y, void* stream);

/* The same step of up to THREE ResBlocks at once -- the three kernel sizes of a HiFi-GAN stage (models.py:457-466:
 * resblock_kernel_sizes 3 / 7 / 11 run side by side on the same input and are averaged) -- as ONE launch: one job per
 * kernel size, all with the same C; each job is what evt_resunit_fwd takes. */
typedef struct evt_resunit_fwd_job {
  evt_resunit_params p;
  const void* x; const void* w1_reg; const void* w2_reg; const float* b1; const float* b2;
  void* xa; void* mid_a; void* y;
} evt_resunit_fwd_job;
int evt_resunit_fwd_multi(const evt_resunit_fwd_job* jobs, int32_t njobs, void* stream);

/* The whole backward of the same step in ONE launch (what torch.autograd runs as two conv backward-data, two conv
 * backward-weight and two bias reductions for modules.py:299-308):
 *   dmid = (c2^T dy') * lrelu'(mid_a),  dx = (c1^T dmid) * lrelu'(xa) + dy',  dy' = dy * dy_scale (rounded to bf16: the
 *   1 / num_kernels of the stage mean, models.py:466, folded into the load; 1.0 = plain dy),
 *   dw2 += dy' (x) mid_a, dw1 += dmid (x) xa (fp32 gradient images, REG geometry), db2 += sum dy', db1 += sum dmid.
 * w1_alt / w2_alt: the ALT images of the two convolutions.  dmid (optional, may be NULL): receives dmid [nseq][L][C].
 * Weight gradients: evt_resunit_bwd_supported(p, 1) says whether the launch can accumulate them (not for C = 32 with
 * 11 taps: the two images do not fit the register file); then dw1, dw2 != NULL and ws = scratch of at least
 * evt_resunit_bwd_ws_floats() floats (one buffer per stream, contents need not survive the call): blocks store partial
 * gradient rows there and a second launch adds them in block order -- the result does not depend on timing.  Otherwise
 * pass dw1 == dw2 == NULL and dmid != NULL and run evt_conv1d_bwd_weight(xa, dmid) / (mid_a, dy) for the two gradients.
 * db1 / db2 may be NULL.  evt_resunit_bwd_multi: one to three jobs (one per kernel size 3 / 7 / 11, same C) in one
 * launch, each job as described. */
int32_t evt_resunit_bwd_supported(const evt_resunit_params* p, int32_t with_weight_grads);
int64_t evt_resunit_bwd_ws_floats(const evt_resunit_params* p);
int evt_resunit_bwd(const evt_resunit_params* p, const void* dy, float dy_scale, const void* xa, const void* mid_a,
                    const void* w1_alt, const void* w2_alt, void* dx, void* dmid, float* dw1, float* dw2, float* db1,
                    float* db2, float* ws, int64_t ws_floats, void* stream);
typedef struct evt_resunit_bwd_job {
  evt_resunit_params p;
  float dy_scale;
  const void* dy; const void* xa; const void* mid_a; const void* w1_alt; const void* w2_alt;
  void* dx; void* dmid;
  float* dw1; float* dw2; float* db1; float* db2;
} evt_resunit_bwd_job;
int evt_resunit_bwd_multi(const evt_resunit_bwd_job* jobs, int32_t njobs, float* ws, int64_t ws_floats, void* stream);

/* The same step for the WIDE vocoder stages (bf16, C in {64, 128}, k in {3, 7, 11}, d in 1..5) with the activated
 * intermediate kept in LDS: forward in one launch (instead of leaky-relu + two convolutions), and the DATA half of the
 * backward in one launch -- dmid = (c2^T dy') * lrelu'(mid_a) and dx = (c1^T dmid) * lrelu'(xa) + dy', dy' = dy * dy_scale
 * rounded to bf16 -- instead of two.  dmid [nseq][L][C] is written for the two weight-gradient launches
 * (evt_conv1d_bwd_weight on (mid_a, dy') and (xa, dmid)), which stay separate: at these widths a gradient image does not
 * fit a wave's registers.  Operands as in evt_resunit_fwd / evt_resunit_bwd. */
int32_t evt_resunit_wide_supported(const evt_resunit_params* p);
int evt_resunit_wide_fwd(const evt_resunit_params* p, const void* x, const void* w1_reg, const void* w2_reg, const float* b1,
                         const float* b2, void* xa, void* mid_a, void* y, void* stream);
int evt_resunit_wide_bwd_data(const evt_resunit_params* p, const void* dy, float dy_scale, const void* xa, const void* mid_a,
                              const void* w1_alt, const void* w2_alt, void* dmid, void* dx, void* stream);

/* One layer of the WN stack of the posterior encoder / the flow (modules.py:187-211 of the reference) FORWARD in one launch
 * (16-bit type, H = 192, k = 5, dilation 1): x_in = in_layer(x) [nseq][L][2H], acts = tanh(x_in[:H] + g[:H]) *
 * sigmoid(x_in[H:] + g[H:]) [nseq][L][H], rs = res_skip(acts); x_out = (x + rs[:H]) * mask, acc_out = acc_in + rs[H:]
 * (last != 0: res_skip has H outputs, acc_out = (acc_in + rs) * mask, x_out unused).  Replaces evt_conv1d_fwd +
 * evt_gated_act_fwd + evt_conv1d_fwd + evt_wn_residual_fwd with the same rounding points (x_in, acts, rs rounded to the
 * 16-bit type); the gate output stays in LDS between the two convolutions.  w_in_frag / w_rs_frag: the two convolutions'
 * REG images (evt_conv1d_layout) re-ordered by evt_frag_pack, b_in [2H] / b_rs [2H or H] fp32 or NULL, g [nseq][2H] or
 * NULL, acc_in NULL for the first layer, lens [nseq] or NULL (mask = position < lens[sequence]).  x_in and acts are outputs
 * the backward launches (evt_gated_act_bwd, the weight gradients) read, as before.  EVT_ENOTSUP outside that shape family.
 *
 * evt_frag_pack: items is a DEVICE table; each item copies a REG or ALT image of `rows` (multiple of 16) rows x `ktot`
 * (multiple of 32) elements of the 16-bit type into FRAGMENT ORDER [rows / 16][ktot / 32][64][8] -- lane (n, g) of (tile, K step)
 * holds row 16 tile + n, elements 32 ks + 8 g .. + 7, i.e. every MFMA A operand is one contiguous 1 KiB load of a wave.
 * One launch per weight fold for all such images of a model (they change when the weights do).  Readers: evt_wn_layer_fwd,
 * evt_wn_layer_bwd_data. */
typedef struct evt_frag_item {
  const void* src;   /* REG or ALT image, [rows][ktot] */
  void* dst;         /* rows * ktot elements */
  int32_t rows, ktot;
} evt_frag_item;
int evt_frag_pack(const evt_frag_item* items, int32_t nitems, void* stream);
/* evt_wn_layer_bwd_data: the data half of the same layer's backward in one launch (instead of evt_wn_residual_bwd, the 1 x 1
 * evt_conv1d_bwd_data, evt_gated_act_bwd and the k = 5 evt_conv1d_bwd_data with its add epilogue):
 *   drs = [dx_next * mask | dacc] [nseq][L][2H]   (last: drs = dacc * mask [nseq][L][H]; dx_next must be NULL)
 *   dacts = res_skip^T drs;  dx_in = gate'(x_in + g) dacts [nseq][L][2H];  dg [nseq][2H] fp32 += sum over positions (or NULL)
 *   dx = in_layer^T dx_in + dx_next * mask [nseq][L][H]
 * w_rs_alt_frag / w_in_alt_frag: the ALT images in fragment order.  drs and dx_in are outputs the two weight-gradient
 * launches (evt_conv1d_bwd_weight on (acts, drs) and (x, dx_in)) read, as before. */
int evt_wn_layer_bwd_data(int32_t dtype, const void* dx_next, const void* dacc, const void* x_in, const void* g,
                          const void* w_rs_alt_frag, const void* w_in_alt_frag, const int32_t* lens, void* drs, void* dx_in,
                          void* dx, float* dg, int32_t nseq, int32_t L, int32_t H, int32_t k, int32_t last, void* stream);
int32_t evt_wn_layer_supported(int32_t dtype, int32_t H, int32_t k, int32_t dil);
int evt_wn_layer_fwd(int32_t dtype, const void* x, const void* w_in_frag, const float* b_in, const void* w_rs_frag,
                     const float* b_rs, const void* g, const void* acc_in, const int32_t* lens, void* x_in, void* acts,
                     void* x_out, void* acc_out, int32_t nseq, int32_t L, int32_t H, int32_t k, int32_t last, void* stream);

/* ---------------------------------------------------------------------------------------
 * Element-wise / reduction helpers of the s2 path.
 * ------------------------------------------------------------------------------------- */
/* out = (a + b + c) * scale ; b, c may be NULL.  (Generator stage mean, models.py:457-466) */
int evt_add3_scale(int32_t dtype, const void* a, const void* b, const void* c, float scale, void* out,
                   int64_t n, void* stream);

/* out = leaky_relu(x, slope).  The HiFi-GAN residual unit (modules.py:299-308) keeps an activated copy of its input
 * so that its convolutions and weight gradients take plain operands (LDS-DMA kernels).  16-byte aligned. */
int evt_leaky_relu(int32_t dtype, const void* x, float slope, void* out, int64_t n, void* stream);

/* out = dy * act'(y), the activation derivative taken through the activation OUTPUT y (leaky-relu keeps the sign,
 * tanh' = 1 - y^2).  Replaces the autograd node of F.leaky_relu after a wide DiscriminatorP conv
 * (models.py:527-531) when evt_conv1d_wants_plain_dy() says the GEMM-grade path will consume dy: both backward
 * GEMMs then read the pre-multiplied tensor instead of fusing the derivative into their loads.  16-byte aligned. */
int evt_dact_mul(int32_t dtype, const void* dy, const void* y, int32_t act_kind, float slope, void* out, int64_t n,
                 void* stream);

/* WN gated activation, commons.py:94-101 (fused_add_tanh_sigmoid_multiply), channels-last:
 *   acts[n][t][h] = tanh(xin[n][t][h] + g[n][h]) * sigmoid(xin[n][t][H+h] + g[n][H+h])
 * g is [nseq][2H] (broadcast over t) or NULL. */
int evt_gated_act_fwd(int32_t dtype, const void* xin, const void* g, void* acts, int32_t nseq, int32_t len,
                      int32_t H, void* stream);
/* dxin = d(acts)/d(xin) * dacts ; dg[nseq][2H] (fp32, +=) may be NULL */
int evt_gated_act_bwd(int32_t dtype, const void* xin, const void* g, const void* dacts, void* dxin, float* dg,
                      int32_t nseq, int32_t len, int32_t H, void* stream);


/* STFT magnitude + mel + log of the generated waveform, src/easevoice/module/mel_processing.py:93-142:
 * reflect-pad (n_fft-hop)/2, hann window, n_fft-point real DFT per frame (radix-2 in LDS, wavefront
 * shuffles for the short butterflies), sqrt(re^2+im^2+1e-6), mel basis [n_mels][n_fft/2+1] matmul,
 * log(clamp(.,1e-5)).  wav is fp32 [nseq][wav_len]; outputs fp32 channels-FIRST [nseq][n_mels][frames]
 * (the layout the mel-L1 at sovits.py:513 consumes).  spec_out ([nseq][n_fft/2+1][frames]) may be NULL. */
int evt_mel_fwd(const float* wav, const float* window, const float* mel_basis, float* spec_out, float* mel_out,
                float* ws_spec_re_im, int32_t nseq, int32_t wav_len, int32_t n_fft, int32_t hop, int32_t n_mels,
                void* stream);
/* dwav[nseq][wav_len] = d(sum(dmel * mel))/d(wav); ws is the (re,im,mag,melpre) workspace written by fwd. */
int evt_mel_bwd(const float* dmel, const float* window, const float* mel_basis, const float* ws_spec_re_im,
                float* dwav, int32_t nseq, int32_t wav_len, int32_t n_fft, int32_t hop, int32_t n_mels, void* stream);
/* workspace floats needed by evt_mel_fwd/bwd */
int64_t evt_mel_workspace_floats(int32_t nseq, int32_t wav_len, int32_t n_fft, int32_t hop, int32_t n_mels);

/* ---------------------------------------------------------------------------------------
 * Input front-end of the s2 step (no gradients).
 * ------------------------------------------------------------------------------------- */
/* Layout change at the model boundary: src fp32 [B][C][T] (the reference's layout: spectrogram, ssl features) ->
 * dst [B][T][Cp] in `dtype`, channels c >= C written as zeros.  With Cp a multiple of 64 the 1025-bin projection
 * enc_q.pre (models.py:348-352) runs on the library's GEMM kernels (its prepared image is padded to Cp columns). */
int evt_ncl_to_nlc(int32_t dtype, const float* src, void* dst, int32_t B, int32_t C, int32_t T, int32_t Cp, void* stream);

/* Frozen quantizer look-up (models.py:912-926 -> quantize.py:70-94 -> core_vq.py:172-228, n_q = 1, eval):
 *   evt_rvq_norms : ee[k] = |embed[k]|^2                                                   (embed fp32 [K][D])
 *   evt_rvq_select: code[n] = argmax_k -(|h_n|^2 - 2 dots[n][k] + ee[k]), first maximum; q_out rows n*rep .. n*rep+rep-1
 *                   = embed[code[n]]  (rep = 2: the x2 nearest up-sampling of the 25 Hz codes)
 * dots [N][K] = h . embed^T comes from evt_conv1d_fwd (a 1x1 layer whose weight is the codebook), h [N][D] from the
 * ssl_proj convolution, both in fp32 whatever the compute dtype of the step (the reference disables autocast there). */
int evt_rvq_norms(const float* embed, float* ee, int32_t K, int32_t D, void* stream);
int evt_rvq_select(const float* h, const float* dots, const float* embed, const float* ee, int64_t* codes, float* q_out,
                   int32_t N, int32_t D, int32_t K, int32_t rep, void* stream);

/* Target-side mel (mel_processing.py:77-90 + commons.slice_segments at sovits.py:478-480):
 *   out[b][m][f] = log(max(sum_k basis[m][k] * spec[b][k][starts[b] + f], 1e-5)),  f < nfr
 * spec fp32 [B][F][T] (reference layout), basis fp32 [M][F], starts int64 [B] or NULL (= 0), out fp32 [B][M][nfr]. */
int evt_spec_to_mel(const float* spec, const float* basis, const int64_t* starts, float* out, int32_t B, int32_t F,
                    int32_t T, int32_t M, int32_t nfr, void* stream);

/* Fused loss reductions (src/easevoice/module/losses.py:7-61).  Pointer tables are DEVICE arrays.
 * feature_loss: out[0] += 2 * sum_i mean|r_i - g_i| ; optional dg_i = 2*sign(g_i-r_i)/n_i * dloss. */
typedef struct evt_seg { const void* a; const void* b; void* da; int64_t n; float scale; int32_t pad_; } evt_seg;
int evt_l1_multi_fwd(int32_t dtype, const evt_seg* segs, int32_t nseg, float* out, void* stream);
int evt_l1_multi_bwd(int32_t dtype, const evt_seg* segs, int32_t nseg, const float* dloss, void* stream);
/* LSGAN: out[0] += sum_i scale_i * mean((target - a_i)^2) ; da_i = -2*scale_i*(target - a_i)/n_i * dloss */
int evt_lsgan_multi_fwd(int32_t dtype, const evt_seg* segs, int32_t nseg, float target, float* out, void* stream);
int evt_lsgan_multi_bwd(int32_t dtype, const evt_seg* segs, int32_t nseg, float target, const float* dloss,
                        void* stream);

/* Masked KL term of the generator loss, src/easevoice/module/losses.py:46-61:
 *   kl = logs_p - logs_q - 0.5 + 0.5 (z_p - m_p)^2 exp(-2 logs_p);  loss = sum(kl * z_mask) / sum(z_mask)
 * with z_mask[b][t] = (t < lens[b]) (lens NULL = all live).  The four tensors share one layout: [B][T][C] (time_inner
 * = 0, channels-last) or [B][C][T] (time_inner = 1, the reference's); each has its own dtype.
 * fwd: out2[0] += sum(kl * z_mask) (caller zeroes it), out2[1] = sum(z_mask) = live frames; loss = out2[0] / out2[1].
 * bwd: gradients of loss * dloss[0] w.r.t. the four inputs (dtype of the input; any may be NULL), masked positions
 * get zeros; `count` points at out2[1]. */
typedef struct evt_kl_params {
  int32_t B, T, C, time_inner;
  int32_t dt_z_p, dt_logs_q, dt_m_p, dt_logs_p;
} evt_kl_params;
int evt_masked_kl_fwd(const evt_kl_params* p, const void* z_p, const void* logs_q, const void* m_p, const void* logs_p,
                      const int32_t* lens, float* out2, void* stream);
int evt_masked_kl_bwd(const evt_kl_params* p, const void* z_p, const void* logs_q, const void* m_p, const void* logs_p,
                      const int32_t* lens, const float* dloss, const float* count, void* dz_p, void* dlogs_q, void* dm_p,
                      void* dlogs_p, void* stream);

/* Multi-segment AdamW over a flat fp32 arena (torch.optim.AdamW at sovits.py:294-319: decoupled
 * weight decay, bias correction, eps outside the sqrt).  seg table is a DEVICE array (<= 64 segments);
 * elements of [0, n) outside every segment are left untouched (frozen parameters, alignment padding). */
typedef struct evt_adamw_seg { int64_t begin, end; float lr; float weight_decay; } evt_adamw_seg;
int evt_adamw_flat(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                   const evt_adamw_seg* segs, int32_t nseg, float beta1, float beta2, float eps, int32_t step,
                   float grad_scale, void* stream);
/* The same update with the step number in DEVICE memory: *step_counter is incremented by one (a 1-thread launch) and
 * the bias corrections are computed from it on the device.  No argument changes from step to step, so the call can be
 * captured into a HIP graph and replayed. */
int evt_adamw_flat_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                       const evt_adamw_seg* segs, int32_t nseg, float beta1, float beta2, float eps,
                       int32_t* step_counter, float grad_scale, void* stream);
/* The same update restricted to the elements [lo, hi) of the arena: lets a step update one sub-model's parameters as soon
 * as its gradients are complete, on a side stream, under the backward of the next sub-model (AdamW is element-wise: the
 * ranges of one step may run in any order and give the bits of the single launch).  bump != 0 increments *step_counter
 * first -- exactly one range call per step passes it, the first one enqueued; all of them must be ordered after it. */
int evt_adamw_flat_dev_range(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t lo, int64_t hi,
                             const evt_adamw_seg* segs, int32_t nseg, float beta1, float beta2, float eps,
                             int32_t* step_counter, int32_t bump, float grad_scale, void* stream);
/* Input side of MultiPeriodDiscriminator (src/easevoice/module/models.py:541-547 DiscriminatorP: reflect-pad to a multiple of
 * the period, view [B, 1, T/p, p]; models.py:576-587 DiscriminatorS: the waveform itself = period 1) for ALL sub-discriminators
 * in one launch.  The waveform batch is [src0 ; src1] (n0 + n1 items of T samples; src1 may be NULL with n1 = 0), for
 * period p_i:   outs[i][(b * p_i + ph)][j][0] = x[b][t],  t = j * p_i + ph,  t >= T mirrored to 2T - 2 - t,
 * outs[i] holding (n0 + n1) * p_i sequences of ceil(T / p_i) positions of one channel, in out_dtype.  nper <= 8; the two
 * sources may differ in dtype (real audio fp32, generated audio in the compute dtype). */
int evt_mpd_fold(int32_t src0_dtype, const void* src0, int32_t n0, int32_t src1_dtype, const void* src1, int32_t n1, int32_t T,
                 const int32_t* periods, int32_t nper, void* const* outs, int32_t out_dtype, void* stream);
/* Its backward: dsrc[b][t] = sum_i (douts[i] at the positions that read x[b0 + b][t], the mirrored tail included), for the
 * n items starting at item b0 of the prepared batches (the generated half of [real ; generated]: b0 = n0). */
int evt_mpd_unfold(int32_t grad_dtype, const void* const* douts, const int32_t* periods, int32_t nper, int32_t b0, int32_t n,
                   int32_t T, int32_t dsrc_dtype, void* dsrc, void* stream);

/* out[0] = sum(x^2) over n floats (grad-norm at commons.py:140-155 without the per-parameter .item()) */
int evt_sumsq(const float* x, int64_t n, float* out, void* stream);

/* ---------------------------------------------------------------------------------------
 * Loss scaling of the fp16 mode (torch.cuda.amp.GradScaler at src/train/sovits.py:378 and :504-507, :521-525:
 * scale(loss).backward(); unscale_(optim); step(optim); update()), with every piece of state in DEVICE memory so that
 * the step stays free of host synchronisation and can be captured into a HIP graph.
 *   scale [1] fp32, growth_tracker [1] int32, found_inf [1] fp32 per optimiser (0 = all gradients finite).
 * evt_scaler_unscale: grad[i] *= 1 / scale (the reciprocal taken in double and rounded to fp32, as GradScaler._unscale_grads_
 *   does), *found_inf = 1 if any element is inf / NaN (torch._amp_foreach_non_finite_check_and_unscale_); found_inf is only
 *   ever raised here, evt_scaler_update clears it.
 * evt_adamw_flat_dev_guarded: evt_adamw_flat_dev that does NOTHING -- no update, no step-counter increment -- when
 *   *skip != 0 (GradScaler.step skips optimizer.step() on an overflow); skip == NULL: unconditional.
 * evt_scaler_update: torch._amp_update_scale_: any of the nflags <= 4 flags set -> scale *= backoff_factor, tracker = 0;
 *   else tracker += 1 and, when it reaches growth_interval, scale *= growth_factor (kept only if finite), tracker = 0.
 *   The flags are reset to 0 for the next step. */
int evt_scaler_unscale(float* grad, int64_t n, const float* scale, float* found_inf, void* stream);
int evt_adamw_flat_dev_guarded(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                               const evt_adamw_seg* segs, int32_t nseg, float beta1, float beta2, float eps,
                               int32_t* step_counter, float grad_scale, const float* skip, void* stream);
int evt_scaler_update(float* scale, int32_t* growth_tracker, float* const* found_inf, int32_t nflags, float growth_factor,
                      float backoff_factor, int32_t growth_interval, void* stream);

/* ---------------------------------------------------------------------------------------
 * s1 (text -> semantic GPT) kernels.
 * ------------------------------------------------------------------------------------- */
/* Dense layers ("gemm_bf16" of SURVEY 8(b)):  y[M][N] = act(x[M][K] . W[N][K]^T + bias),  act = relu when `relu`.
 * Replaces F.linear at transformer.py:207-224,330-334 (linear1 / linear2), patched_mha_with_cache.py:242,460 (packed
 * in-projection, out-projection) and t2s_model.py:276,486 (bert_proj, ar_predict_layer).  dtype EVT_DT_BF16 runs on the
 * LDS-DMA MFMA kernels (fp32 accumulation), EVT_DT_F32 is the fp32-MFMA parity path.  K % 8 == 0 and N % 8 == 0 (a
 * caller with an odd N, e.g. the 1025-entry vocabulary, pads the weight image and the output to a multiple of 128 with
 * zero rows / columns).
 * The weights are passed as the two prepared images of evt_gemm_bf16_layout (same structs as the convolutions:
 * evt_wn_fold_multi with g == NULL casts an fp32 master into both images in one launch for all layers):
 *   REG = W row-major [N][K] in chunks of reg_ck (forward, dW geometry),  ALT = W^T [K][N] (backward-data).
 * bwd_data: dx[M][K] = dy[M][N] . W   (dy already multiplied by the activation derivative: evt_dact_mul)
 * bwd_weight: dw (fp32, REG geometry) += dy^T x ; dbias[N] (fp32, may be NULL) += column sums of dy. */
typedef struct evt_gemm_params {
  int32_t dtype;   /* EVT_DT_* */
  int32_t M, N, K;
  int32_t relu;    /* forward epilogue: max(0, .) */
} evt_gemm_params;
int evt_gemm_bf16_layout(const evt_gemm_params* g, evt_wlayout* out);
int evt_gemm_bf16_fwd(const evt_gemm_params* g, const void* x, const void* w_reg, const void* w_alt, const float* bias,
                      void* y, void* stream);
int evt_gemm_bf16_bwd_data(const evt_gemm_params* g, const void* dy, const void* w_reg, const void* w_alt, void* dx,
                           void* stream);
int evt_gemm_bf16_bwd_weight(const evt_gemm_params* g, const void* x, const void* dy, float* dw, float* dbias,
                             void* stream);

/* The same GEMMs with fused epilogues, on the 256 x 256 x 64 kernel of csrc/gemm256.hip (bf16, N and K multiples of 256 /
 * 64 on the output / reduction side, M >= 2048: the shapes of the s1 blocks at training batch sizes).  What the s1 block
 * (transformer.py:311-334) otherwise runs as separate element-wise launches rides in the GEMM's store:
 *   forward       y  = dropout(act(x . W^T + bias)) [+ add]     mask = counter hash of (*seed_dev, site, row * N + col),
 *                                                                the one evt_relu_dropout_fwd draws
 *   backward-data dx = (dy . W) * (gate > 0 ? gate_pos : 0) [+ add]
 *                 gate = the saved dropout(relu(.)) output of the layer below (its sign pattern IS the relu + dropout
 *                 derivative, gate_pos = 1 / (1 - p)); add = the residual branch's gradient.
 * evt_gemm_bf16_fused_supported says whether a shape is covered (callers fall back to the plain entry points plus the
 * element-wise launches); the _ex entry points return EVT_ENOTSUP otherwise.  epi may be NULL (plain GEMM). */
typedef struct evt_gemm_epilogue {
  float dropout_p;          /* 0 = off */
  uint32_t site;
  const uint32_t* seed_dev; /* device counter of the dropout stream (may be NULL = 0) */
  const void* gate;         /* [M][out columns] in the GEMM's dtype, or NULL */
  float gate_pos;
  uint32_t pad_;
  const void* add;          /* [M][out columns] in the GEMM's dtype, or NULL; added last */
} evt_gemm_epilogue;
int32_t evt_gemm_bf16_fused_supported(const evt_gemm_params* g, int32_t backward_data);
/* measurement switch of csrc/gemm256.hip (tools/bench_gemm256.py): ablation variants of the kernel, bit 0 no MFMAs, bit 1
 * no DMA after the prologue, bit 2 no fragment reads, bit 3 no stores; 0 = the product kernel.  Results are only
 * meaningful for 0. */
void evt_debug_gemm256_variant(int32_t variant);
int evt_gemm_bf16_fwd_ex(const evt_gemm_params* g, const void* x, const void* w_reg, const void* w_alt, const float* bias,
                         const evt_gemm_epilogue* epi, void* y, void* stream);
int evt_gemm_bf16_bwd_data_ex(const evt_gemm_params* g, const void* dy, const void* w_reg, const void* w_alt,
                              const evt_gemm_epilogue* epi, void* dx, void* stream);

/* Flash attention with the ANALYTIC prefix-LM + key-padding mask of
 * src/easevoice/soundstorm/auto_reg/models/t2s_model.py:456-479 (no [B*H,L,L] mask tensor):
 *   key j visible from query i  <=>  j is not padding  AND  ( j < x_len  if i < x_len  else  j <= i )
 * where padding keys are x-part columns j >= x_lens[b] and y-part columns j - x_len >= y_lens[b].
 * q,k,v,o: [B][L][H][D] (the packed in_proj output viewed per head), D = 32 or 64, bf16 or fp32 storage;
 * softmax scale = 1/sqrt(D); lse: fp32 [B][H][L] (log-sum-exp, saved for backward).
 * Replaces F.scaled_dot_product_attention at patched_mha_with_cache.py:452-454. */
typedef struct evt_attn_params {
  int32_t dtype, B, L, H, D;
  int32_t x_len;             /* padded text length (prefix width) */
  int64_t q_stride_b, q_stride_l, q_stride_h; /* element strides (q, k, v share them) */
  int64_t o_stride_b, o_stride_l, o_stride_h;
  float dropout_p;           /* attention dropout on the normalised probabilities (SDPA dropout_p); 0 = off */
  uint32_t seed;             /* mask = hash(seed, b*H+h, query, key): identical in forward and backward */
} evt_attn_params;
int evt_attn_prefixlm_fwd(const evt_attn_params* p, const void* q, const void* k, const void* v,
                          const int32_t* x_lens, const int32_t* y_lens, void* o, float* lse, void* stream);
int evt_attn_prefixlm_bwd(const evt_attn_params* p, const void* q, const void* k, const void* v, const void* o,
                          const void* d_o, const float* lse, const int32_t* x_lens, const int32_t* y_lens,
                          void* dq, void* dk, void* dv, float* delta_ws, void* stream);

/* y = LayerNorm(x + r) * gamma + beta over the last dim C (post-LN block, transformer.py:311-315);
 * r may be NULL.  mean/rstd fp32 [rows] are saved for backward. */
int evt_add_layernorm_fwd(int32_t dtype, const void* x, const void* r, const float* gamma, const float* beta,
                          void* y, float* mean, float* rstd, int64_t rows, int32_t C, float eps, void* stream);
/* dxr = d/d(x+r); dgamma/dbeta fp32 [C] are accumulated (+=) */
int evt_add_layernorm_bwd(int32_t dtype, const void* x, const void* r, const float* gamma, const void* dy,
                          const float* mean, const float* rstd, void* dxr, float* dgamma, float* dbeta,
                          int64_t rows, int32_t C, void* stream);

/* ---------------------------------------------------------------------------------------
 * s2 transformer encoder glue (enc_p), src/easevoice/module/attentions.py:60-75.
 * ------------------------------------------------------------------------------------- */
/* *counter += inc, a 1-thread launch.  Device-side step / RNG counters: nothing about them is a host argument, so the
 * launches that read them can be replayed from a captured HIP graph. */
int evt_counter_inc(uint32_t* counter, uint32_t inc, void* stream);

/* out = LayerNorm(x + dropout(y)) * gamma + beta, rows (b, t) with t >= lens[b] written as zeros (lens may be NULL).
 * x, y, out [rows][C] in `dtype`; gamma/beta fp32 [C]; mean/rstd fp32 [rows] saved for the backward.
 * dropout keep(element) = hash(*seed_dev, site, element index) >= p * 2^32, kept values scaled by 1/(1-p); the backward
 * regenerates the mask from the same (seed, site).  rows_per_seq = T (row = b*T + t). */
int evt_res_dropout_ln_fwd(int32_t dtype, const void* x, const void* y, const float* gamma, const float* beta,
                           const int32_t* lens, int32_t rows_per_seq, float p, const uint32_t* seed_dev, uint32_t site,
                           void* out, float* mean, float* rstd, int64_t rows, int32_t C, float eps, void* stream);
/* dx = d(loss)/dx, dy = d(loss)/dy (= dx * dropout multiplier; may be NULL when p == 0: then dy == dx);
 * dgamma/dbeta fp32 [C] are accumulated (+=). */
int evt_res_dropout_ln_bwd(int32_t dtype, const void* x, const void* y, const float* gamma, const void* dout,
                           const float* mean, const float* rstd, const int32_t* lens, int32_t rows_per_seq, float p,
                           const uint32_t* seed_dev, uint32_t site, void* dx, void* dy, float* dgamma, float* dbeta,
                           int64_t rows, int32_t C, void* stream);

/* y = dropout(relu(x)) / dx = dy * dropout multiplier * (x > 0): the FFN inner activation of the s1 transformer blocks
 * (src/easevoice/soundstorm/auto_reg/modules/transformer.py:330-334), one pass each way; mask from the counter hash
 * (seed_dev, site, element index) as in evt_res_dropout_ln_*.  n % 8 == 0 (bf16) / % 4 (fp32), 16-byte aligned.
 * With lens != NULL the tensor is [.., rows_per_seq, C] and rows t >= lens[b] are zeroed (the `* x_mask` of the s2
 * encoder FFN, attentions.py:408-416); lens == NULL: no mask. */
int evt_relu_dropout_fwd(int32_t dtype, const void* x, float p, const uint32_t* seed_dev, uint32_t site,
                         const int32_t* lens, int32_t rows_per_seq, int32_t C, void* y, int64_t n, void* stream);
int evt_relu_dropout_bwd(int32_t dtype, const void* x, const void* dy, float p, const uint32_t* seed_dev, uint32_t site,
                         const int32_t* lens, int32_t rows_per_seq, int32_t C, void* dx, int64_t n, void* stream);

/* WN layer glue (src/easevoice/module/modules.py:199-211): rs = res_skip_layer(acts), [rows][2H] (H when last):
 *   last == 0:  x_out = (x + rs[:, :H]) * row_mask ;  acc_out = acc + rs[:, H:]     (acc may be NULL = zeros)
 *   last == 1:  acc_out = (acc + rs) * row_mask
 * backward: drs[:, :H] = dx = dx_out * row_mask, drs[:, H:] = dacc_out (last: drs = dacc_out * row_mask); d(acc) is
 * dacc_out itself and d(x) is dx.  dx_out / dacc_out may be NULL (= zeros). */
int evt_wn_residual_fwd(int32_t dtype, const void* x, const void* rs, const void* acc, const int32_t* lens,
                        int32_t rows_per_seq, void* x_out, void* acc_out, int64_t rows, int32_t H, int32_t last,
                        void* stream);
/* Element-wise chains of the style encoder (MelStyleEncoder, src/easevoice/module/modules.py:685-763), dropout masks from the
 * device counter + `site` as in evt_relu_dropout_*:
 *   Mish + Dropout (modules.py:521-545 LinearNorm / Mish / Dropout of `spectral`):  y = drop(x * tanh(softplus(x)))
 *   Conv1dGLU tail (modules.py:548-566):  y = res + drop(h[:, :C] * sigmoid(h[:, C:])),  h [rows][2C], res / y [rows][C]
 * backward: dx = dy * mask/(1-p) * mish'(x);  dh from dy (the residual's gradient is dy itself).
 * dtype: x / h and their gradients; wide_dtype: y / res / dy -- the same, or EVT_DT_F32 next to bf16 operands (the
 * reference's autocast keeps Mish outputs and the GLU residual stream in fp32 around its half-precision projections). */
int evt_mish_dropout_fwd(int32_t dtype, int32_t wide_dtype, const void* x, float p, const uint32_t* seed_dev, uint32_t site,
                         void* y, int64_t n, void* stream);
int evt_mish_dropout_bwd(int32_t dtype, int32_t wide_dtype, const void* x, const void* dy, float p, const uint32_t* seed_dev,
                         uint32_t site, void* dx, int64_t n, void* stream);
int evt_glu_dropout_res_fwd(int32_t dtype, int32_t wide_dtype, const void* h, const void* res, float p,
                            const uint32_t* seed_dev, uint32_t site, void* y, int64_t rows, int32_t C, void* stream);
int evt_glu_dropout_res_bwd(int32_t dtype, int32_t wide_dtype, const void* h, const void* dy, float p,
                            const uint32_t* seed_dev, uint32_t site, void* dh, int64_t rows, int32_t C, void* stream);
/* Tail of the posterior encoder (src/easevoice/module/models.py:352-358): with stats [rows][2C] the projection's output,
 *   m = stats[:, :C] * mask, logs = stats[:, C:] * mask, z = (m + eps * exp(logs)) * mask     (fp32 [rows][C] each)
 * and its backward: dstats from dz / dm / dlogs (each may be NULL), eps and the saved logs. */
int evt_reparam_fwd(int32_t dtype, const void* stats, const float* eps, const int32_t* lens, int32_t rows_per_seq,
                    int64_t rows, int32_t C, float* z, float* m, float* logs, void* stream);
int evt_reparam_bwd(int32_t dtype, const float* dz, const float* dm, const float* dlogs, const float* eps, const float* logs,
                    const int32_t* lens, int32_t rows_per_seq, int64_t rows, int32_t C, void* dstats, void* stream);
/* Mean-only residual coupling + Flip of the s2 flow (src/easevoice/module/modules.py:404-458 forward with logs == 0 (mean_only),
 * models.py:273-315: flows = [coupling, Flip] x 4), everything after the layer's `post` projection in one launch:
 *   y[row][c] = v[row][2h-1-c],  v = [ x[:, :h] , (x[:, h:] + stats) * row_mask ];  x0n = y[:, :h] in `dtype` (the next
 *   layer's projection input) or NULL.  x, y fp32 [rows][2h]; stats [rows][h] in `dtype`; the row mask from lens as above. */
int evt_coupling_flip_fwd(int32_t dtype, const float* x, const void* stats, const int32_t* lens, int32_t rows_per_seq,
                          int64_t rows, int32_t h, float* y, void* x0n, void* stream);
/* dy [rows][2h] fp32 (+ dx0n [rows][h] in `dtype`, the gradient that arrived at x0n; may be NULL) ->
 * dx [rows][2h] fp32, dstats [rows][h] in `dtype` */
int evt_coupling_flip_bwd(int32_t dtype, const float* dy, const void* dx0n, const int32_t* lens, int32_t rows_per_seq,
                          int64_t rows, int32_t h, float* dx, void* dstats, void* stream);
int evt_wn_residual_bwd(int32_t dtype, const void* dx_out, const void* dacc_out, const int32_t* lens,
                        int32_t rows_per_seq, void* dx, void* drs, int64_t rows, int32_t H, int32_t last, void* stream);

/* Multi-head attention core of the s2 encoders: the windowed relative-position SELF-attention of enc_p
 * (attentions.py:214-292, window_size = w), the window-less CROSS-attention of MRTE (mrte_model.py:25-61 -> the same
 * MultiHeadAttention.attention with window_size = None, queries = ssl frames, keys / values = phonemes) and the style
 * encoder's ScaledDotProductAttention (modules.py:605-682, scale = 1/sqrt(d_model)):
 *   scores[i][j] = scale * (q_i . k_j + [window >= 0 and |j-i| <= w] q_i . Ek[j-i+w]);  keys j >= lens_k[b] excluded
 *   p = dropout(softmax_j(scores));   out_i = sum_j p[i][j] (v_j + [window >= 0 and |j-i| <= w] Ev[j-i+w])
 * q: rows [B][Tq][ldq], k, v: rows [B][Tk][ldk], head h in columns [h*D, (h+1)*D) (slices of one packed projection are
 * fine); out [B][Tq][ldo] likewise, all in `dtype` (bf16: MFMA flash kernels, D % 32 == 0; fp32: the 1e-3 parity path,
 * D % 4 == 0; D <= 128).  emb_k / emb_v fp32 [n_heads_rel][2w+1][D] (NULL when window < 0); lse fp32 [B*H][Tq] is saved
 * for the backward.  lens_q / lens_k int32 [B] or NULL (= all rows live).  Query rows i >= lens_q[b] are written as
 * zeros.  Dropout mask = hash(*seed_dev, site, b, h, i, j), regenerated by the backward. */
typedef struct evt_mha_params {
  int32_t dtype;           /* EVT_DT_* of q, k, v, out and their gradients */
  int32_t B, H, D;
  int32_t Tq, Tk;          /* query rows / key rows per item */
  int32_t window;          /* w >= 0: relative positions (needs Tq == Tk, 2w+1 <= 16); < 0: none */
  int32_t n_heads_rel;     /* 1 (heads share the embeddings) or H */
  int64_t ldq, ldk, ldo;   /* row strides in elements; 16-byte multiples */
  float scale;             /* logit scale: 1/sqrt(D) (attentions.py:257), 1/sqrt(d_model) (modules.py:618) */
  float dropout_p;
  uint32_t site;
  uint32_t pad_;
  const uint32_t* seed_dev;
} evt_mha_params;
int evt_mha_fwd(const evt_mha_params* p, const void* q, const void* k, const void* v, const float* emb_k,
                const float* emb_v, const int32_t* lens_q, const int32_t* lens_k, void* out, float* lse, void* stream);
/* dq (stride ldq), dk, dv (stride ldk) in `dtype`; demb_k / demb_v fp32, accumulated (+=), NULL when window < 0;
 * delta_ws fp32 [B*H][Tq] scratch. */
int evt_mha_bwd(const evt_mha_params* p, const void* q, const void* k, const void* v, const void* o, const void* d_o,
                const float* lse, const float* emb_k, const float* emb_v, const int32_t* lens_q, const int32_t* lens_k,
                void* dq, void* dk, void* dv, float* demb_k, float* demb_v, float* delta_ws, void* stream);

/* Cross-entropy, reduction="sum" (t2s_model.py:486-489): logits [rows][V] (any dtype), targets int64.
 * loss[0] += sum_r (lse_r - logit[r][t_r]); dlogits = (softmax - onehot) * dloss[0]; top-k hit counts for
 * the accuracy metric are written to hits[0] (+=, rows with target == ignore_index skipped, count in hits[1]). */
int evt_ce_sum_fwd_bwd(int32_t dtype, const void* logits, const int64_t* targets, void* dlogits, float* loss,
                       int32_t* hits, int64_t rows, int32_t V, int32_t topk, int64_t ignore_index, float dloss,
                       void* stream);
/* Per-row form for the DPO branch (t2s_model.py:420-427, models/utils.py:176-183): row_loss[r] = lse_r - logit[r][t_r]
 * (= -log p(target), written, not accumulated), dlogits = softmax - onehot (unscaled; the caller applies the per-row
 * upstream weight), hits as above.  The summed target log-probability of a sequence is -sum of its rows. */
int evt_ce_rows_fwd_bwd(int32_t dtype, const void* logits, const int64_t* targets, void* dlogits, float* row_loss,
                        int32_t* hits, int64_t rows, int32_t V, int32_t topk, int64_t ignore_index, void* stream);
/* The same two entry points for logits / dlogits with a row stride ld >= V (the GEMM above writes the 1025-entry
 * vocabulary into rows of 1152): columns [V, ld) of the logits are ignored and written as zeros in dlogits. */
int evt_ce_sum_fwd_bwd_ld(int32_t dtype, const void* logits, const int64_t* targets, void* dlogits, float* loss,
                          int32_t* hits, int64_t rows, int32_t V, int64_t ld, int32_t topk, int64_t ignore_index,
                          float dloss, void* stream);
int evt_ce_rows_fwd_bwd_ld(int32_t dtype, const void* logits, const int64_t* targets, void* dlogits, float* row_loss,
                           int32_t* hits, int64_t rows, int32_t V, int64_t ld, int32_t topk, int64_t ignore_index,
                           void* stream);

/* ---------------------------------------------------------------------------------------
 * s1 KV-cache decoding step (Text2SemanticDecoder.infer_panel_naive, t2s_model.py:762-863).
 * Every per-step quantity is device state, so the launches of a step take the same arguments for every token and can be
 * captured into a HIP graph.  ctr is a device int32[8]:
 * ------------------------------------------------------------------------------------- */
#define EVT_DEC_POS 0    /* key/value positions already in the cache */
#define EVT_DEC_IDX 1    /* decode step index (`idx` of t2s_model.py:822) */
#define EVT_DEC_YCOUNT 2 /* tokens in the y buffer (prompt + generated) */
#define EVT_DEC_YLEN 3   /* prompt length: the new token's position is YLEN + IDX (t2s_model.py:861) */
#define EVT_DEC_SEED 4   /* per-call sampling seed, xor-ed with evt_sample_params.seed */
/* y[b][n] = act(bias[n] + sum_k W[n][k] * x[b][k]); W [N][K] row-major in `wdtype`, vectors fp32, B <= 4, K % 512 == 0.
 * r == NULL: x = a.  r != NULL: x = LayerNorm(a + r) * ln_g + ln_b (the post-LN residual of T2SBlock.decode_next_token,
 * t2s_model.py:208-221, recomputed per workgroup); x_out (may be NULL) receives it for the next residual and must not
 * alias a or r.  relu != 0 applies max(0, .) (T2SMLP.forward, t2s_model.py:74-77). */
int evt_dec_gemv(int32_t wdtype, const void* W, const float* bias, const float* a, const float* r, const float* ln_g,
                 const float* ln_b, float ln_eps, float* x_out, float* y, int32_t B, int32_t N, int32_t K, int32_t relu,
                 void* stream);
/* qkv fp32 [B][3*H*D] of the new token: its key/value are stored at cache position ctr[POS] (kcache/vcache
 * [B][Lmax][H*D] in `cdtype`) and out[b][h*D+d] = softmax(q.K^T/sqrt(D)).V over positions 0..ctr[POS]
 * (t2s_model.py:187-204 without the torch.cat growth).  D == 32.
 * x_lens (may be NULL): int32 [B]; cache positions x_lens[b] <= j < x_len are the padding of a shorter text in a batch
 * (the padding mask of infer_panel_batch_infer, t2s_model.py:620-650) and are not attended. */
int evt_dec_attn(int32_t cdtype, const float* qkv, void* kcache, void* vcache, const int32_t* ctr, float* out, int32_t B,
                 int32_t H, int32_t D, int32_t Lmax, const int32_t* x_lens, int32_t x_len, void* stream);
/* evt_dec_gemv (the packed in-projection) and evt_dec_attn in one launch: workgroup (b, h) computes only its head's
 * 3*D rows of Wqkv [3*H*D][H*D] (+ bqkv) from x = a or LayerNorm(a + r) (workgroups h == 0 store it to x_out), then
 * appends and attends as evt_dec_attn does.  The weights and the cache share `dtype`.  H*D == 512, D == 32. */
int evt_dec_qkv_attn(int32_t dtype, const void* Wqkv, const float* bqkv, const float* a, const float* r,
                     const float* ln_g, const float* ln_b, float ln_eps, float* x_out, void* kcache, void* vcache,
                     const int32_t* ctr, float* out, int32_t B, int32_t H, int32_t D, int32_t Lmax, const int32_t* x_lens,
                     int32_t x_len, void* stream);
/* sample() of models/utils.py:125-171 for one step: the EOS column is dropped while ctr[IDX] < no_eos_steps
 * (t2s_model.py:833-834); repetition penalty over y[b][0..ctr[YCOUNT]); nucleus cut (top_p < 1) on the un-tempered
 * logits; division by max(temperature, 1e-5); top-k pivot (top_k <= 0: off; ties kept); softmax;
 * token = argmax(probs / q) with q = noise[ctr[IDX]][v] (noise fp32 [steps][V]) or, when noise == NULL, exponential
 * noise from a counter hash of (seed ^ ctr[SEED], step, b, v).  The token is written to y[b][ctr[YCOUNT]] (y int64 [B][ymax]);
 * stop_idx[b] (int32, initialised to -1 by the caller) receives the first step at which argmax of the penalised logits
 * or the token equals eos (t2s_model.py:846).  probs_out (may be NULL) fp32 [B][V]. */
typedef struct evt_sample_params {
  int32_t V, eos, top_k, no_eos_steps, ymax;
  float top_p, temperature, repetition_penalty;
  uint32_t seed;
  int32_t noise_rows;   /* 1: noise [steps][V] shared by the batch rows; B: noise [steps][B][V] */
} evt_sample_params;
int evt_dec_sample(const evt_sample_params* p, const float* logits, int64_t* y, const int32_t* ctr, const float* noise,
                   int32_t* stop_idx, float* probs_out, int32_t B, void* stream);
/* evt_dec_sample + evt_dec_embed + evt_dec_advance(dpos) for ONE sequence in a single launch */
int evt_dec_sample_embed(const evt_sample_params* p, const float* logits, int64_t* y, int32_t* ctr, const float* noise,
                         int32_t* stop_idx, const float* emb, const float* pe, const float* alpha, float x_scale, float* x,
                         int32_t E, int32_t npos, int32_t dpos, void* stream);
/* x[b] = emb[y[b][ctr[YCOUNT]]] * x_scale + alpha[0] * pe[ctr[YLEN] + ctr[IDX]]  (t2s_model.py:860-861); emb fp32
 * [V][E], pe fp32 [npos][E]. */
int evt_dec_embed(const float* emb, const float* pe, const float* alpha, float x_scale, const int64_t* y,
                  const int32_t* ctr, float* x, int32_t B, int32_t E, int32_t ymax, int32_t npos, void* stream);
/* end of a step: ctr[IDX] += 1, ctr[YCOUNT] += 1, ctr[POS] += dpos (1 after a decode step, 0 after the prompt pass) */
int evt_dec_advance(int32_t* ctr, int32_t dpos, void* stream);

/* ScaledAdam (src/easevoice/soundstorm/auto_reg/modules/optim.py:206-251,300-390,448-622) over a flat fp32 arena.
 * The reference stacks same-shaped tensors only to batch its torch ops; the arithmetic is per tensor, which is what
 * these two launches implement for ALL tensors at once:
 *   evt_scaled_adam_stats : stats[t] = (sum p*g, sum p*p, sum g*g) per tensor t  (+=, caller zeroes stats)
 *   evt_scaled_adam_apply : delta = beta1*delta + p*coef[t][0] + g/(sqrt(v_hat)+eps)*coef[t][1] ; p += delta
 *                           (coef[t][2] != 0 selects the numel==1 "scalar" branch, optim.py:600-622).  As in the
 *                           reference, gradient clipping only enters through coef[t][0] (the size update).
 * The per-tensor scalars between the two launches (size update, clipping scale) are a few vector ops on [ntensor]
 * device arrays done by the host layer.  chunks: DEVICE table, one workgroup per chunk (a slice of one tensor). */
typedef struct evt_sa_chunk { int64_t begin, end; int32_t tensor; int32_t pad_; } evt_sa_chunk;
typedef struct evt_scaled_adam_hp {
  float lr, beta1, beta2, eps, scalar_lr_scale, scalar_max;
  int32_t step, pad_;
} evt_scaled_adam_hp;
int evt_scaled_adam_stats(const float* param, const float* grad, const evt_sa_chunk* chunks, int32_t nchunks,
                          float* stats, void* stream);
int evt_scaled_adam_apply(float* param, const float* grad, float* delta, float* exp_avg_sq, const evt_sa_chunk* chunks,
                          int32_t nchunks, const float* coef, const evt_scaled_adam_hp* hp, void* stream);

/* ---------------------------------------------------------------------------------------
 * Feature extractors behind the training set (src/normalization/normalize.py:88-106 `_get_bert_feature`,
 * :132-180 `_name2go` of the reference: transformers' BertForMaskedLM / HubertModel on the CPU).  Their transformer
 * layers run on the entry points above (1x1 convolutions = the Linear layers, evt_mha_fwd, evt_res_dropout_ln_fwd);
 * these two are what they need besides.  Forward only (inference).
 *   evt_gelu_rows_fwd: out[b][t][:] = gelu(x[b][t][:] + bias) for t < t_out <= t_in, erf form (transformers' "gelu");
 *     x [nseq][t_in][C], out [nseq][t_out][C], bias fp32 [C] or NULL.  t_out = t_in - 1 is the trim of HuBERT's positional
 *     convolution (HubertSamePadLayer drops the last frame of an even kernel).
 *   evt_channel_norm_gelu_fwd: per (sequence, channel) normalisation over the T frames, affine, then GELU if apply_gelu:
 *     GroupNorm(num_groups = C) + GELU of HuBERT's first feature-extractor layer.  x, out [nseq][T][C]; biased variance. */
int evt_gelu_rows_fwd(int32_t dtype, const void* x, const float* bias, void* out, int64_t nseq, int32_t t_in, int32_t t_out,
                      int32_t C, void* stream);
int evt_channel_norm_gelu_fwd(int32_t dtype, const void* x, const float* gamma, const float* beta, float eps, void* out,
                              int32_t nseq, int32_t T, int32_t C, int32_t apply_gelu, void* stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* EVT_H */
