"""Fixture of the s1 `precision: 16-mixed` mode (configs/gpt.yaml:6; SURVEY section 8(f) N4, VERDICT r5 missing #2): the
REFERENCE's own Text2SemanticDecoder.forward_old, ScaledAdam and WarmupCosineLRSchedule run through the body of
Text2SemanticLightningModule.training_step (t2s_lightning_module.py:41-56, restated below) the way Lightning's
mixed-precision plugin runs it under `16-mixed` with manual optimisation:

    forward under torch.autocast(dtype=float16)            (MixedPrecision.forward_context)
    manual_backward(loss) = scaler.scale(loss).backward()  (MixedPrecision.pre_backward)
    opt.step()            = scaler.step(opt); scaler.update()   (MixedPrecision.optimizer_step; scaler.step unscales the
                                                                 gradients once, skips the optimiser on inf / nan)
    opt.zero_grad(); scheduler.step()                      on `batch_idx > 0 and batch_idx % 4 == 0`

with torch's own autocast (device "cpu": matrix products in IEEE half), torch.amp.GradScaler and the reference optimiser.
Lightning itself is not installable here and is not needed: the three plugin hooks above are its whole contribution.
Build container only (/root/reference is imported read-only through oracle/refshim.py); tests/test_s1_fp16_gpu.py reads
the .pt this writes.

Thirteen micro-batches = three optimiser windows (batch_idx 0..4: FIVE backward passes precede the first step, then 5..8
and 9..12 -- the losses of the third window are those of the weights ScaledAdam wrote at the end of the second).
The scaler is built with init_scale = 2**40, backoff_factor = 2**-32, growth_interval = 2 (the arithmetic of scale /
unscale / skip / update does not depend on the constants; these visit every branch with wide margins): window 1 overflows
(a SUM-reduced cross-entropy scaled by 2**40 cannot be differentiated in half precision) -> the step is skipped, the
gradients are dropped, the scale backs off to 2**8 (2**12 still overflows in the reference under CPU autocast); windows 2 and 3 are clean -> ScaledAdam steps twice at 2**8 (growth tracker 1, then 2), and the scale grows to 2**9 after the
second clean step.  (With growth_interval = 1 the third window runs at 2**9 and overflows in the reference: the loss
has dropped by then and a half-precision gradient inside the stack passes 65504 -- a margin a fixture should not sit on.)

  python tests/golden/make_golden_s1_fp16.py
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import refshim  # noqa: E402

refshim.install()
sys.path.insert(0, os.path.join(ROOT, "tests"))
from util_fill import fill_module, s1_batch  # noqa: E402

SCALER = dict(init_scale=2.0 ** 40, backoff_factor=2.0 ** -32, growth_factor=2.0, growth_interval=2)
SLICES = ["bert_proj.weight", "ar_text_embedding.word_embeddings.weight", "ar_audio_embedding.word_embeddings.weight",
          "ar_audio_position.alpha", "h.layers.0.self_attn.in_proj_weight", "h.layers.0.self_attn.out_proj.weight",
          "h.layers.11.linear1.weight", "h.layers.23.linear2.bias", "h.layers.23.norm2.weight", "ar_predict_layer.weight"]
B, X_LEN, Y_LEN = 2, 64, 192
X_LENS, Y_LENS = [64, 41], [192, 150]          # ragged: padded keys and padded targets are on the path


def batches():
    """two alternating micro-batches (different token draws), lengths ragged"""
    out = []
    for seed in (1234, 4321):
        b = s1_batch(B, X_LEN, Y_LEN, seed=seed)
        b["phoneme_ids_len"] = torch.tensor(X_LENS)
        b["semantic_ids_len"] = torch.tensor(Y_LENS)
        out.append(b)
    return out


def top_of(name):
    return ".".join(name.split(".")[:3]) if name.startswith("h.layers") else name.split(".")[0]


def main():
    import yaml
    from src.easevoice.soundstorm.auto_reg.models.t2s_model import Text2SemanticDecoder
    from src.easevoice.soundstorm.auto_reg.modules.lr_schedulers import WarmupCosineLRSchedule
    from src.easevoice.soundstorm.auto_reg.modules.optim import ScaledAdam

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
    names = [n for n, _ in model.named_parameters()]
    params = dict(model.named_parameters())
    # configure_optimizers (t2s_lightning_module.py:94-122)
    opt = ScaledAdam(model.parameters(), lr=0.01, betas=(0.9, 0.95), clipping_scale=2.0, parameters_names=[names],
                     show_dominant_parameters=False, clipping_update_period=1000)
    o = cfg["optimizer"]
    sched = WarmupCosineLRSchedule(opt, init_lr=o["lr_init"], peak_lr=o["lr"], end_lr=o["lr_end"],
                                   warmup_steps=o["warmup_steps"], total_steps=o["decay_steps"])
    scaler = torch.amp.GradScaler("cpu", **SCALER)
    p0 = {n: params[n].detach().clone() for n in SLICES}
    bs = batches()
    rec = dict(losses=[], accs=[], scale_before=[], stepped=[], skipped=[], scale_after=[], tracker_after=[], windows=[])
    for batch_idx in range(13):
        b = bs[batch_idx % 2]
        rec["scale_before"].append(float(scaler.get_scale()))
        with torch.autocast("cpu", dtype=torch.float16):
            loss, acc = model.forward_old(b["phoneme_ids"], b["phoneme_ids_len"], b["semantic_ids"], b["semantic_ids_len"],
                                          b["bert_feature"])
        scaler.scale(loss).backward()                                   # manual_backward under the AMP plugin
        rec["losses"].append(float(loss))
        rec["accs"].append(float(acc))
        stepped = batch_idx > 0 and batch_idx % 4 == 0
        rec["stepped"].append(stepped)
        if stepped:
            scaler.unscale_(opt)                                        # what scaler.step does first; done here to record
            finite = all(bool(torch.isfinite(p.grad).all()) for p in model.parameters() if p.grad is not None)
            w = dict(batch_idx=batch_idx, finite=finite)
            if finite:
                gss = {}
                for n, p in params.items():
                    gss[top_of(n)] = gss.get(top_of(n), 0.0) + float(p.grad.double().pow(2).sum())
                w["grad_sumsq"] = gss
                w["grad_slices"] = {n: params[n].grad.flatten()[:96].clone() for n in SLICES}
                w["grad_abs_sum"] = {n: float(params[n].grad.double().abs().sum()) for n in SLICES}
            before = {n: params[n].detach().clone() for n in SLICES}
            scaler.step(opt)                                            # skips opt.step() when a gradient is not finite
            scaler.update()
            opt.zero_grad()
            sched.step()
            w["moved"] = any(not torch.equal(before[n], params[n].detach()) for n in SLICES)
            w["param_slices"] = {n: params[n].detach().flatten()[:96].clone() for n in SLICES}
            w["delta_abs_sum"] = {n: float((params[n].detach() - before[n]).double().abs().sum()) for n in SLICES}
            rec["windows"].append(w)
            rec["skipped"].append(not w["moved"])
        rec["scale_after"].append(float(scaler.get_scale()))
        rec["tracker_after"].append(int(scaler.state_dict()["_growth_tracker"]))
        print(batch_idx, "loss", rec["losses"][-1], "acc", rec["accs"][-1], "scale", rec["scale_before"][-1], "->",
              rec["scale_after"][-1], "stepped", stepped, flush=True)
    out = dict(config=dict(B=B, x_len=X_LEN, y_len=Y_LEN, x_lens=X_LENS, y_lens=Y_LENS, batch_seeds=[1234, 4321],
                           scaler=SCALER, fill_seed=3, micro_batches=13),
               init_slices={n: p0[n].flatten()[:96].clone() for n in SLICES}, **rec)
    assert rec["skipped"] == [True, False, False], rec["skipped"]
    assert rec["scale_after"][4] == 2.0 ** 8 and rec["scale_after"][8] == 2.0 ** 8 and rec["scale_after"][12] == 2.0 ** 9
    assert [rec["tracker_after"][i] for i in (4, 8, 12)] == [0, 1, 0]
    path = os.path.join(HERE, "s1_fp16.pt")
    torch.save(out, path)
    print("wrote", path, "losses", [round(x, 3) for x in rec["losses"]])


if __name__ == "__main__":
    main()
