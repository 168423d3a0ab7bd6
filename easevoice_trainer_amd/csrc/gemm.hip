// evt_gemm_bf16_*: the dense-layer GEMMs of the s1 transformer (and any 1x1 projection) on the hand-written MFMA
// kernels of this library (gfx950 only).
//
// Reference call sites: F.linear of TransformerEncoderLayer.linear1 / linear2
// (src/easevoice/soundstorm/auto_reg/modules/transformer.py:207-224,330-334), the packed in-projection and the
// out-projection of the attention (patched_mha_with_cache.py:242,460), bert_proj and ar_predict_layer
// (models/t2s_model.py:276,296,486).
//
// y[M][N] = act(x[M][K] . W[N][K]^T + bias) is the k = 1 member of the implicit-GEMM convolution family with the M rows
// as positions of one sequence: the same LDS-DMA kernels run it (conv_deep / conv_deep32: 128 x 128 tile,
// global_load_lds_dwordx4 staging with the XOR swizzle on the source address, mfma_f32_16x16x32_bf16; conv_ring for
// 64-wide tails; wgrad_deep / wgrad_ring with ds_read_b64_tr_b16 fragments for dW = dy^T x), chosen by the same
// dispatcher.  What is specific to a GEMM is only the descriptor, built here.  fp32 (the parity path) runs on
// conv_igemm<float> (mfma_f32_16x16x4f32).
#include "evt_common.h"
#include "../../include/evt.h"

namespace {

int make(const evt_gemm_params* g, evt_conv1d_params* c, int out_act) {
  if (!g || g->M <= 0 || g->N <= 0 || g->K <= 0) return EVT_EINVAL;
  if (g->dtype != EVT_DT_HALF && g->dtype != EVT_DT_F32) return EVT_EINVAL;
  if (g->K % 8 || g->N % 8) return EVT_ENOTSUP;          // 16-byte rows (callers pad N, e.g. the 1025-wide vocabulary)
  c->dtype = g->dtype;
  c->nseq = 1;
  c->lin = g->M;
  c->cin = g->K;
  c->cout = g->N;
  c->k = 1; c->stride = 1; c->pad = 0; c->dil = 1; c->groups = 1; c->transposed = 0;
  c->in_slope = 1.f;
  c->out_act = out_act;
  c->out_slope = 0.f;                                     // relu = leaky-relu with slope 0
  c->impl = EVT_IMPL_AUTO;
  return EVT_OK;
}

}  // namespace

extern "C" {

int evt_gemm_bf16_layout(const evt_gemm_params* g, evt_wlayout* out) {
  evt_conv1d_params c;
  if (int rc = make(g, &c, EVT_ACT_NONE)) return rc;
  return evt_conv1d_layout(&c, out);
}

int evt_gemm_bf16_fwd(const evt_gemm_params* g, const void* x, const void* w_reg, const void* w_alt, const float* bias,
                      void* y, void* stream) {
  evt_conv1d_params c;
  if (int rc = make(g, &c, g && g->relu ? EVT_ACT_LRELU : EVT_ACT_NONE)) return rc;
  return evt_conv1d_fwd(&c, x, w_reg, w_alt, bias, nullptr, y, stream);
}

int evt_gemm_bf16_bwd_data(const evt_gemm_params* g, const void* dy, const void* w_reg, const void* w_alt, void* dx,
                           void* stream) {
  evt_conv1d_params c;
  if (int rc = make(g, &c, EVT_ACT_NONE)) return rc;
  return evt_conv1d_bwd_data(&c, dy, nullptr, w_reg, w_alt, nullptr, nullptr, dx, stream);
}

int evt_gemm_bf16_bwd_weight(const evt_gemm_params* g, const void* x, const void* dy, float* dw, float* dbias,
                             void* stream) {
  evt_conv1d_params c;
  if (int rc = make(g, &c, EVT_ACT_NONE)) return rc;
  return evt_conv1d_bwd_weight(&c, x, dy, nullptr, dw, dbias, stream);
}

}  // extern "C"
