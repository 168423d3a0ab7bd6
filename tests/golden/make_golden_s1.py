"""s1 fixtures from the REFERENCE's own Text2SemanticDecoder / ScaledAdam (build container only)."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import refshim  # noqa: E402

refshim.install()
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)
from util_fill import fill_module, s1_batch  # noqa: E402


def make_s1():
    import yaml
    from src.easevoice.soundstorm.auto_reg.models.t2s_model import Text2SemanticDecoder
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
    B, x_len, y_len = 3, 24, 40
    b = s1_batch(B, x_len, y_len)
    b["phoneme_ids_len"] = torch.tensor([24, 17, 9])
    b["semantic_ids_len"] = torch.tensor([40, 29, 33])
    loss, acc = model.forward_old(b["phoneme_ids"], b["phoneme_ids_len"], b["semantic_ids"], b["semantic_ids_len"],
                                  b["bert_feature"])
    model.zero_grad()
    loss.backward()
    names = ["bert_proj.weight", "ar_text_embedding.word_embeddings.weight", "ar_text_position.alpha",
             "ar_audio_embedding.word_embeddings.weight", "ar_audio_position.alpha", "h.layers.0.self_attn.in_proj_weight",
             "h.layers.0.self_attn.in_proj_bias", "h.layers.0.self_attn.out_proj.weight", "h.layers.11.linear1.weight",
             "h.layers.23.linear2.bias", "h.layers.23.norm2.weight", "h.layers.5.norm1.bias", "ar_predict_layer.weight"]
    params = dict(model.named_parameters())
    grads = {n: params[n].grad.flatten()[:96].clone() for n in names}
    gss = {}
    for n, p in params.items():
        top = ".".join(n.split(".")[:3]) if n.startswith("h.layers") else n.split(".")[0]
        gss[top] = gss.get(top, 0.0) + float(p.grad.double().pow(2).sum())
    out = dict(config=dict(B=B, x_len=x_len, y_len=y_len, x_lens=[24, 17, 9], y_lens=[40, 29, 33]),
               loss=float(loss), acc=float(acc), grad_slices=grads, grad_sumsq=gss)

    # ---- ScaledAdam trajectory on a small parameter set (covers batching by shape, the scalar branch, the size update
    #      every 4 steps and clipping with a short update period) ----
    g = torch.Generator().manual_seed(5)
    shapes = dict(w1=(8, 16), w2=(8, 16), b=(8,), s=(1,), e=(5, 7), s2=(1,))
    ps = {k: torch.nn.Parameter(torch.randn(v, generator=g) * (0.5 if len(v) > 1 else 0.2)) for k, v in shapes.items()}
    opt = ScaledAdam(list(ps.values()), lr=0.01, betas=(0.9, 0.95), clipping_scale=2.0, clipping_update_period=4,
                     parameters_names=[list(ps.keys())], show_dominant_parameters=False)
    traj, grads_used = [], []
    for step in range(14):
        gs = {}
        for k, p in ps.items():
            scale = 5.0 if step in (9, 12) else 1.0       # spikes so that clipping engages
            p.grad = torch.randn(p.shape, generator=g) * 0.1 * scale
            gs[k] = p.grad.clone()
        grads_used.append(gs)
        opt.step()
        for gr in opt.param_groups:
            gr["lr"] = 0.002                                # what WarmupCosineLRSchedule.step() does after every step
        traj.append({k: p.detach().clone() for k, p in ps.items()})
    out["scaled_adam"] = dict(shapes=shapes, init={k: v for k, v in zip(ps.keys(), [None] * len(ps))},
                              grads=grads_used, traj=traj)
    g2 = torch.Generator().manual_seed(5)
    out["scaled_adam"]["init"] = {k: torch.randn(v, generator=g2) * (0.5 if len(v) > 1 else 0.2) for k, v in shapes.items()}
    path = os.path.join(HERE, "s1_small.pt")
    torch.save(out, path)
    print("wrote", path, "loss", out["loss"], "acc", out["acc"], "per-token nll", out["loss"] / (B * y_len))


def make_s1_dpo():
    """the DPO branch (Text2SemanticDecoder.forward, t2s_model.py:393-429): the rejected sequences come from
    make_reject_y's draws on torch's global CPU generator, seeded here; they are stored so the product / oracle can be
    checked both on the draw and on the arithmetic."""
    import yaml
    from src.easevoice.soundstorm.auto_reg.models import t2s_model as TM
    from src.easevoice.soundstorm.auto_reg.models.utils import make_reject_y

    torch.set_num_threads(8)
    cfg = yaml.safe_load(open(os.path.join(refshim.REFERENCE_ROOT, "configs", "gpt.yaml")))
    model = TM.Text2SemanticDecoder(config=cfg, top_k=3)
    fill_module(model, 3)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if hasattr(m, "dropout") and isinstance(m.dropout, float):
            m.dropout = 0.0
    model.train()
    B, x_len, y_len = 3, 24, 40
    b = s1_batch(B, x_len, y_len)
    x_lens, y_lens = torch.tensor([24, 17, 9]), torch.tensor([40, 29, 33])
    cases = []
    for seed in (2024, 558):          # 558: the rejected sequences are only 1/0/1 tokens longer -> the DPO term matters
        torch.manual_seed(seed)
        reject_y, reject_y_lens = make_reject_y(b["semantic_ids"], y_lens)
        torch.manual_seed(seed)
        loss, acc = model.forward(b["phoneme_ids"], x_lens, b["semantic_ids"], y_lens, b["bert_feature"])
        model.zero_grad()
        loss.backward()
        names = ["bert_proj.weight", "ar_text_embedding.word_embeddings.weight", "ar_audio_embedding.word_embeddings.weight",
                 "ar_audio_position.alpha", "h.layers.0.self_attn.in_proj_weight", "h.layers.11.linear1.weight",
                 "h.layers.23.linear2.bias", "h.layers.23.norm2.weight", "ar_predict_layer.weight"]
        params = dict(model.named_parameters())
        gss = {}
        for n, p in params.items():
            top = ".".join(n.split(".")[:3]) if n.startswith("h.layers") else n.split(".")[0]
            gss[top] = gss.get(top, 0.0) + float(p.grad.double().pow(2).sum())
        # the two parts of the loss, recomputed with the reference's helpers on the same rejected batch
        from src.easevoice.soundstorm.auto_reg.models.utils import dpo_loss, get_batch_logps
        with torch.no_grad():
            xy, mask, targets = model.make_input_data(b["phoneme_ids"], x_lens, b["semantic_ids"], y_lens, b["bert_feature"])
            logits = model.ar_predict_layer(model.h((xy, None), mask=mask)[0][:, x_len:])
            rxy, rmask, rtargets = model.make_input_data(b["phoneme_ids"], x_lens, reject_y, reject_y_lens, b["bert_feature"])
            rlogits = model.ar_predict_layer(model.h((rxy, None), mask=rmask)[0][:, x_len:])
            a_lp, r_lp = get_batch_logps(logits, rlogits, targets, rtargets)
            loss_2 = dpo_loss(a_lp, r_lp, 0, 0, 0.2, reference_free=True)[0]
        out = dict(config=dict(B=B, x_len=x_len, y_len=y_len, x_lens=[24, 17, 9], y_lens=[40, 29, 33], seed=seed),
                   reject_y=reject_y, reject_y_lens=reject_y_lens, loss=float(loss), acc=float(acc),
                   chosen_logps=a_lp, rejected_logps=r_lp, loss_dpo=float(loss_2),
                   grad_slices={n: params[n].grad.flatten()[:96].clone() for n in names}, grad_sumsq=gss)
        cases.append(out)
        print("seed", seed, "loss", out["loss"], "dpo part", out["loss_dpo"], "acc", out["acc"], "reject lens",
              reject_y_lens.tolist())
    path = os.path.join(HERE, "s1_dpo.pt")
    torch.save(dict(cases=cases), path)
    print("wrote", path)


from make_golden_s1_inputs import infer_inputs  # noqa: E402


INFER_CASES = [dict(top_k=15, top_p=1, temperature=1.0, repetition_penalty=1.35, early_stop_num=40, prompt=True),
               dict(top_k=5, top_p=0.8, temperature=0.7, repetition_penalty=1.2, early_stop_num=25, prompt=True),
               dict(top_k=15, top_p=1, temperature=1.0, repetition_penalty=1.35, early_stop_num=20, prompt=False)]


def make_s1_infer():
    """KV-cache decoding (Text2SemanticDecoder.infer_panel_naive, t2s_model.py:762-863) of the reference's own module.
    The only stand-in: the exponential noise of multinomial_sample_one_no_sync (models/utils.py:118-122) comes from a
    seeded table q[step] instead of the global generator, so the token sequence is reproducible on any device."""
    import yaml
    from src.easevoice.soundstorm.auto_reg.models import t2s_model as TM
    from src.easevoice.soundstorm.auto_reg.models import utils as U

    torch.set_num_threads(8)
    cfg = yaml.safe_load(open(os.path.join(refshim.REFERENCE_ROOT, "configs", "gpt.yaml")))
    model = TM.Text2SemanticDecoder(config=cfg, top_k=3)
    fill_module(model, 3)
    model.eval()
    d = infer_inputs()
    state = dict(step=0, logits=[])

    def sample_one(probs):
        qrow = d["q"][state["step"], :probs.size(-1)]
        state["step"] += 1
        return torch.argmax(probs / qrow, dim=-1, keepdim=True).to(dtype=torch.int)

    orig_one, orig_l2p = U.multinomial_sample_one_no_sync, U.logits_to_probs

    def l2p(logits, previous_tokens=None, **kw):
        state["logits"].append(logits.detach().clone())        # before the in-place repetition penalty
        return orig_l2p(logits=logits, previous_tokens=previous_tokens, **kw)

    U.multinomial_sample_one_no_sync, U.logits_to_probs = sample_one, l2p
    cases = []
    try:
        with torch.no_grad():
            for c in INFER_CASES:
                state["step"], state["logits"] = 0, []
                kw = {k: v for k, v in c.items() if k != "prompt"}
                y, idx = model.infer_panel_naive(d["x"], torch.tensor([24]), d["prompts"] if c["prompt"] else None,
                                                 d["bert"], **kw)
                cases.append(dict(args=c, y=y.clone(), idx=int(idx), steps=state["step"],
                                  logits0=state["logits"][0][0].clone(), logits7=state["logits"][7][0].clone(),
                                  logits_last=state["logits"][-1][0].clone()))
                print(c, "->", tuple(y.shape), "idx", idx, "steps", state["step"], "tail", y[0, -6:].tolist())
    finally:
        U.multinomial_sample_one_no_sync, U.logits_to_probs = orig_one, orig_l2p
    torch.save(dict(cases=cases), os.path.join(HERE, "s1_infer.pt"))


BATCH_CASES = [dict(rows=[0, 1, 2], top_k=1100, top_p=1, temperature=1.0, repetition_penalty=1.35, early_stop_num=20),
               dict(rows=[1], top_k=15, top_p=1, temperature=1.0, repetition_penalty=1.35, early_stop_num=12)]


def make_s1_batch_infer():
    """the default decoding path of the reference's TTS (parallel_infer=True): infer_panel_batch_infer,
    t2s_model.py:563-730, with padded texts, per-row EOS stops and the early stop.  Stand-in as in make_s1_infer: the
    exponential noise comes from a seeded table, here one row per batch item."""
    import yaml
    from make_golden_s1_inputs import batch_infer_inputs
    from src.easevoice.soundstorm.auto_reg.models import t2s_model as TM
    from src.easevoice.soundstorm.auto_reg.models import utils as U

    torch.set_num_threads(8)
    cfg = yaml.safe_load(open(os.path.join(refshim.REFERENCE_ROOT, "configs", "gpt.yaml")))
    model = TM.Text2SemanticDecoder(config=cfg, top_k=3)
    fill_module(model, 3)
    model.eval()
    d = batch_infer_inputs()
    state = dict(step=0, rows=None)

    def sample_one(probs):
        qrow = d["q"][state["step"]][state["rows"][:probs.size(0)], :probs.size(-1)]
        state["step"] += 1
        return torch.argmax(probs / qrow, dim=-1, keepdim=True).to(dtype=torch.int)

    orig = U.multinomial_sample_one_no_sync
    U.multinomial_sample_one_no_sync = sample_one
    cases = []
    try:
        with torch.no_grad():
            for c in BATCH_CASES:
                rows = c["rows"]
                state["step"], state["rows"] = 0, rows
                kw = {k: v for k, v in c.items() if k != "rows"}
                x_lens = d["x_lens"][rows]
                ys, idxs = model.infer_panel_batch_infer([d["x"][r] for r in rows], x_lens, d["prompts"][rows],
                                                         [d["bert"][r] for r in rows], max_len=int(x_lens.max()), **kw)
                cases.append(dict(args=c, y=[y.clone() for y in ys], idx=[int(i) for i in idxs], steps=state["step"]))
                print(c, "-> lens", [int(y.numel()) for y in ys], "idx", idxs, "steps", state["step"])
    finally:
        U.multinomial_sample_one_no_sync = orig
    torch.save(dict(cases=cases), os.path.join(HERE, "s1_batch_infer.pt"))


def make_pipeline():
    """the model-side core of TTS.run (tts.py:756-817) chained from the reference's own s1 and s2 modules: batched
    decoding of two fragments, then SynthesizerTrn.decode over the concatenation (speed 1.0) and per fragment (1.25)"""
    import json
    import math
    import yaml
    from make_golden_s1_inputs import pipeline_inputs
    from util_fill import decode_inputs
    from src.easevoice.module import models as RM
    from src.easevoice.soundstorm.auto_reg.models import t2s_model as TM
    from src.easevoice.soundstorm.auto_reg.models import utils as U

    torch.set_num_threads(8)
    cfg = yaml.safe_load(open(os.path.join(refshim.REFERENCE_ROOT, "configs", "gpt.yaml")))
    t2s = TM.Text2SemanticDecoder(config=cfg, top_k=3)
    fill_module(t2s, 3)
    t2s.eval()
    hps = json.load(open(os.path.join(refshim.REFERENCE_ROOT, "configs", "s2.json")))
    vits = RM.SynthesizerTrn(1025, 32, n_speakers=300, **hps["model"])
    fill_module(vits, 1)
    vits.eval()
    d, dd = pipeline_inputs(), decode_inputs()
    state = dict(step=0)

    def sample_one(probs):
        qrow = d["q"][state["step"], :probs.size(0), :probs.size(-1)]
        state["step"] += 1
        return torch.argmax(probs / qrow, dim=-1, keepdim=True).to(dtype=torch.int)

    orig_one, orig_randn = U.multinomial_sample_one_no_sync, torch.randn_like
    U.multinomial_sample_one_no_sync = sample_one
    out = {}
    try:
        with torch.no_grad():
            lens = torch.tensor([int(i.numel()) for i in d["all_ids"]])
            pred, idxs = t2s.infer_panel_batch_infer(d["all_ids"], lens, d["prompt"].expand(2, -1), d["bert"], top_k=1100,
                                                     top_p=1, temperature=1.0, early_stop_num=50 * cfg["data"]["max_sec"],
                                                     max_len=int(lens.max()), repetition_penalty=1.35)
            out["pred"], out["idx"] = [p.clone() for p in pred], [int(i) for i in idxs]
            torch.randn_like = lambda t, **kw: dd["noise"][:, :, :t.size(2)].to(t.dtype)
            cut = [p[-i:] for p, i in zip(pred, idxs)]
            up = math.prod(vits.upsample_rates)
            ends = [0]
            for p in cut:
                ends.append(ends[-1] + p.shape[0] * 2 * up)
            audio = vits.decode(torch.cat(cut)[None, None], torch.cat(d["batch_phones"])[None], dd["refers"], speed=1.0)[0, 0]
            out["speed1"] = [audio[ends[i - 1]:ends[i]].clone() for i in range(1, len(ends))]
            out["speed125"] = [vits.decode(p[-i:][None, None], ph[None], dd["refers"], speed=1.25)[0, 0].clone()
                               for p, i, ph in zip(pred, idxs, d["batch_phones"])]
    finally:
        U.multinomial_sample_one_no_sync, torch.randn_like = orig_one, orig_randn
    print("tokens", [int(p.numel()) for p in out["pred"]], "idx", out["idx"], "audio", [int(a.numel()) for a in out["speed1"]],
          [int(a.numel()) for a in out["speed125"]])
    torch.save(out, os.path.join(HERE, "pipeline.pt"))


if __name__ == "__main__":
    if "pipeline" in sys.argv[1:]:
        make_pipeline()
    elif "batch" in sys.argv[1:]:
        make_s1_batch_infer()
    elif "infer" in sys.argv[1:]:
        make_s1_infer()
    elif "dpo" in sys.argv[1:]:
        make_s1_dpo()
    else:
        make_s1()
