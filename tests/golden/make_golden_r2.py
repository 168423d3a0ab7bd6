"""Round-2 fixtures, again from the REFERENCE's own modules (build container only; /root/reference is imported
read-only through oracle/refshim.py):

  s2_c1_adamw.pt  the C1 run of make_golden.py continued by ONE optimiser step of both networks with the reference's
                  torch.optim.AdamW set-up (src/train/sovits.py:285-319: four generator groups, three at
                  text_low_lr_rate; betas/eps of configs/s2.json) -- gradients and post-step values of selected
                  parameters, and fp64 checksums of every top-level module's post-step weights
  s2_c2.pt        BASELINE config 2 shapes (B = 16, 4 s clips, text 60), fp32: the seven loss terms, a strided sample of
                  y_hat, y_hat_mel, gradient sums per module and gradient slices
  s1_c3.pt        BASELINE config 3 sequence shape (256 phonemes + 768 semantic tokens) at B = 2 (the CPU reference needs
                  ~6.5 GB per item at L = 1024): loss, top-3 accuracy, gradient sums per block, gradient slices

  s1_c3_b4.pt     the same at B = 4 with four different (text, semantic) length pairs

  python tests/golden/make_golden_r2.py [adamw] [c2] [s1c3] [s1c3b4]
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, HERE]
import make_golden as MG  # noqa: E402  (installs the shims)
from make_golden import refshim  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tests"))
from util_fill import fill_module, s1_batch  # noqa: E402

ADAMW_G = ["enc_p.text_embedding.weight", "enc_p.encoder_text.attn_layers.0.conv_q.bias",
           "enc_p.encoder_text.ffn_layers.1.conv_2.bias", "enc_p.mrte.c_post.weight", "enc_p.mrte.c_pre.bias",
           "enc_p.encoder2.norm_layers_1.0.gamma", "dec.conv_post.weight", "dec.resblocks.14.convs1.0.weight_v",
           "dec.ups.4.weight_g", "dec.cond.bias", "flow.flows.0.enc.in_layers.0.bias", "enc_q.enc.res_skip_layers.15.weight_g",
           "ref_enc.fc.fc.bias"]
ADAMW_D = ["discriminators.0.convs.0.weight_v", "discriminators.3.conv_post.bias", "discriminators.1.convs.0.weight_g",
           "discriminators.5.convs.1.weight_v"]


def g_groups(net_g, lr, low):
    """the four groups of src/train/sovits.py:285-313, built the way the reference builds them (by parameter identity)"""
    te_p = list(map(id, net_g.enc_p.text_embedding.parameters()))
    et_p = list(map(id, net_g.enc_p.encoder_text.parameters()))
    mrte_p = list(map(id, net_g.enc_p.mrte.parameters()))
    base = filter(lambda p: id(p) not in te_p + et_p + mrte_p and p.requires_grad, net_g.parameters())
    return [{"params": base, "lr": lr},
            {"params": net_g.enc_p.text_embedding.parameters(), "lr": low},
            {"params": net_g.enc_p.encoder_text.parameters(), "lr": low},
            {"params": net_g.enc_p.mrte.parameters(), "lr": low}]


def checksums(module):
    out = {}
    for n, p in module.named_parameters():
        top = n.split(".")[0]
        s, q = out.get(top, (0.0, 0.0))
        out[top] = (s + float(p.detach().double().sum()), q + float(p.detach().double().pow(2).sum()))
    return out


def make_adamw():
    cfg = json.load(open(os.path.join(refshim.REFERENCE_ROOT, "configs", "s2.json")))
    t = cfg["train"]
    lr, low = t["learning_rate"], t["learning_rate"] * t["text_low_lr_rate"]
    res = {}

    def hook_d(net_d):
        # what optim_d.step() consumes in the training loop (sovits.py:504-507).  The generator pass of the fixture runs
        # through the UN-stepped D (make_golden.py) and, as in the reference, adds its own gradients to D's .grad, which
        # the loop throws away at the next zero_grad: the D gradients are put back before the step below.
        res["_d_all"] = {n: p.grad.clone() for n, p in net_d.named_parameters()}

    def hook(net_g, net_d, out):
        for n, p in net_d.named_parameters():
            p.grad.copy_(res["_d_all"][n])
        del res["_d_all"]
        opt_g = torch.optim.AdamW(g_groups(net_g, lr, low), lr, betas=t["betas"], eps=t["eps"])
        opt_d = torch.optim.AdamW(net_d.parameters(), lr, betas=t["betas"], eps=t["eps"])
        pg, pd = dict(net_g.named_parameters()), dict(net_d.named_parameters())
        res["grads_g"] = {n: pg[n].grad.clone() for n in ADAMW_G}
        res["grads_d"] = {n: pd[n].grad.clone() for n in ADAMW_D}
        # (the pre-step values are util_fill.fill_tensor(name, shape, seed 1 / 2): not stored)
        res["grad_sumsq_g"] = float(sum(p.grad.double().pow(2).sum() for p in net_g.parameters() if p.grad is not None))
        res["grad_sumsq_d"] = float(sum(p.grad.double().pow(2).sum() for p in net_d.parameters() if p.grad is not None))
        opt_d.step()
        opt_g.step()
        res["after_g"] = {n: pg[n].detach().clone() for n in ADAMW_G}
        res["after_d"] = {n: pd[n].detach().clone() for n in ADAMW_D}
        res["checksums_g"], res["checksums_d"] = checksums(net_g), checksums(net_d)
        res["ssl_proj_unchanged"] = bool(torch.equal(pg["ssl_proj.weight"].detach(),
                                                     MG.fill_module.__globals__["fill_tensor"]("ssl_proj.weight", pg["ssl_proj.weight"].shape, 1)))
        res["hyper"] = dict(lr=lr, low=low, betas=t["betas"], eps=t["eps"], weight_decay=0.01)
        res["losses"] = out["losses"]

    MG.make_s2(tag="_noise", hook=hook, hook_d=hook_d)
    path = os.path.join(HERE, "s2_c1_adamw.pt")
    torch.save(res, path)
    print("wrote", path, "ssl_proj unchanged:", res["ssl_proj_unchanged"], "|g|^2", res["grad_sumsq_g"], res["grad_sumsq_d"])


def make_c2():
    keep = {}

    def hook(net_g, net_d, out):
        keep.update(out)

    MG.make_s2(B=16, T=200, t_text=60, tag="_noise", hook=hook)
    # the reference's own fp32 noise floor at this shape: same computation, one thread, no oneDNN (other summation order)
    nt, mk = torch.get_num_threads(), torch.backends.mkldnn.enabled
    torch.set_num_threads(1)
    torch.backends.mkldnn.enabled = False
    try:
        alt = MG.make_s2(B=16, T=200, t_text=60, tag="_noise")
    finally:
        torch.set_num_threads(nt)
        torch.backends.mkldnn.enabled = mk
    relerr = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-12))
    noise = dict(g_slices={n: relerr(alt["g_grad_slices"][n], v) for n, v in keep["g_grad_slices"].items()},
                 d_slices={n: relerr(alt["d_grad_slices"][n], v) for n, v in keep["d_grad_slices"].items()},
                 g_sumsq={k: abs(alt["g_grad_sumsq"][k] - v) / v for k, v in keep["g_grad_sumsq"].items()},
                 d_sumsq={k: abs(alt["d_grad_sumsq"][k] - v) / v for k, v in keep["d_grad_sumsq"].items()},
                 losses={k: abs(alt["losses"][k] - v) / max(abs(v), 1e-9) for k, v in keep["losses"].items()},
                 y_hat=relerr(alt["y_hat"], keep["y_hat"]))
    out = dict(noise=noise, config=keep["config"], losses=keep["losses"], y_hat_strided=keep["y_hat"][:, ::997].clone(),
               y_hat_rms=float(keep["y_hat"].double().pow(2).mean().sqrt()),
               y_hat_mel_strided=keep["y_hat_mel"][:, ::7, ::3].clone(), y_mel_strided=keep["y_mel"][:, ::7, ::3].clone(),
               d_logits_head=[t[:, :16].clone() for t in keep["d_logits"]],
               stats={k: v.clone() for k, v in keep["stats"].items()},
               g_grad_sumsq=keep["g_grad_sumsq"], d_grad_sumsq=keep["d_grad_sumsq"],
               g_grad_slices=keep["g_grad_slices"], d_grad_slices=keep["d_grad_slices"])
    path = os.path.join(HERE, "s2_c2.pt")
    torch.save(out, path)
    print("wrote", path, {k: round(v, 5) for k, v in out["losses"].items()})


def make_s1c3(B=2, x_lens=(256, 201), y_lens=(768, 645), seed=4321, name="s1_c3.pt"):
    import yaml
    from src.easevoice.soundstorm.auto_reg.models.t2s_model import Text2SemanticDecoder

    torch.set_num_threads(8)
    cfg = yaml.safe_load(open(os.path.join(refshim.REFERENCE_ROOT, "configs", "gpt.yaml")))
    model = Text2SemanticDecoder(config=cfg, top_k=3)
    fill_module(model, 3)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, "dropout") and isinstance(m.dropout, float):
            m.dropout = 0.0
    model.train()
    x_len, y_len = 256, 768
    b = s1_batch(B, x_len, y_len, seed=seed)
    x_lens, y_lens = list(x_lens), list(y_lens)      # s1_c3.pt: one full item, one padded on both sides
    b["phoneme_ids_len"], b["semantic_ids_len"] = torch.tensor(x_lens), torch.tensor(y_lens)
    # the two position scales are single scalars whose gradient is a sum of +- terms over every (token, channel):
    # d alpha = sum(grad_out * pe).  The size of the terms, sum(|grad_out * pe|), is recorded next to the sums so that the
    # comparison can be made the way a dot product's error is bounded -- relative to the magnitudes, not to a result that
    # cancellation may have made arbitrarily small
    abs_terms = {}

    def watch(name):
        mod = getattr(model, name)

        def fwd_hook(m, inp, out):
            pe = m.pe[:, : inp[0].size(1)].detach()
            out.register_hook(lambda g, pe=pe: abs_terms.__setitem__(name, float((g.double() * pe.double()).abs().sum())))
        return mod.register_forward_hook(fwd_hook)

    handles = [watch("ar_text_position"), watch("ar_audio_position")]
    loss, acc = model.forward_old(b["phoneme_ids"], b["phoneme_ids_len"], b["semantic_ids"], b["semantic_ids_len"],
                                  b["bert_feature"])
    model.zero_grad()
    loss.backward()
    for h in handles:
        h.remove()
    names = ["bert_proj.weight", "ar_text_embedding.word_embeddings.weight", "ar_text_position.alpha",
             "ar_audio_embedding.word_embeddings.weight", "ar_audio_position.alpha", "h.layers.0.self_attn.in_proj_weight",
             "h.layers.0.self_attn.in_proj_bias", "h.layers.0.self_attn.out_proj.weight", "h.layers.11.linear1.weight",
             "h.layers.23.linear2.bias", "h.layers.23.norm2.weight", "h.layers.23.self_attn.in_proj_weight",
             "h.layers.5.norm1.bias", "ar_predict_layer.weight"]
    params = dict(model.named_parameters())
    grads = {n: params[n].grad.flatten()[:96].clone() for n in names}
    gss = {}
    for n, p in params.items():
        top = ".".join(n.split(".")[:3]) if n.startswith("h.layers") else n.split(".")[0]
        gss[top] = gss.get(top, 0.0) + float(p.grad.double().pow(2).sum())
    out = dict(config=dict(B=B, x_len=x_len, y_len=y_len, x_lens=x_lens, y_lens=y_lens, seed=seed), loss=float(loss),
               acc=float(acc), grad_slices=grads, grad_sumsq=gss, alpha_abs_terms=abs_terms)
    path = os.path.join(HERE, name)
    torch.save(out, path)
    print("wrote", path, "loss", out["loss"], "acc", out["acc"], "per-token nll", out["loss"] / (B * y_len))


if __name__ == "__main__":
    what = sys.argv[1:] or ["adamw", "c2", "s1c3"]
    if "adamw" in what:
        make_adamw()
    if "c2" in what:
        make_c2()
    if "s1c3" in what:
        make_s1c3()
    if "s1c3b4" in what:
        # four ragged items (~26 GB of host memory in the reference): multi-block grids per (batch, head) beyond two items,
        # a text shorter than half the padded length, a semantic sequence shorter than half
        make_s1c3(B=4, x_lens=(256, 201, 130, 256), y_lens=(768, 645, 768, 300), seed=4322, name="s1_c3_b4.pt")
