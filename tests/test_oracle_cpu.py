"""CPU: the oracle restatement (oracle/s2_step.py) is pinned against fixtures generated from the REFERENCE's own
modules (tests/golden/s2_c1.pt).  This is what makes the oracle trustworthy as the checker for arbitrary sizes."""
import json
import sys
import os

import torch

from oracle import s2_step as O
from util_fill import fill_tensor, s2_batch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def rel(a, b):
    return ((a.detach().float() - b.float()).abs().max() / (b.float().abs().max() + 1e-12)).item()


def _filled(keys, seed):
    return {k: fill_tensor(k, shape, seed) for k, shape in keys.items()}


def test_oracle_s2_step_matches_reference_fixture():
    torch.set_num_threads(8)
    gold = torch.load(os.path.join(HERE, "golden", "s2_c1.pt"), weights_only=False)
    keys = json.load(open(os.path.join(HERE, "golden", "state_dict_keys.json")))
    hps = json.load(open(os.path.join(ROOT, "configs", "s2.json")))
    sd_g, sd_d = _filled(keys["s2_g"], 1), _filled(keys["s2_d"], 2)
    c = gold["config"]
    b = s2_batch(c["B"], c["T"], c["t_text"])
    out = O.s2_losses(sd_g, sd_d, hps, b["ssl"], b["wav"], b["text"], b["lengths"], b["text_lengths"], b["eps"],
                      b["ids_slice"], with_grads=True)
    assert rel(out["spec"][:, :, :4], gold["spec_head"]) < 1e-5
    assert rel(out["y_hat"].squeeze(1), gold["y_hat"]) < 1e-4
    assert rel(out["y_hat_mel"], gold["y_hat_mel"]) < 1e-4
    for k in ("disc", "gen", "fm", "mel", "kl", "gen_all"):
        assert abs(float(out[k]) - gold["losses"][k]) <= 2e-4 * abs(gold["losses"][k]), (k, float(out[k]))
    for n, s in gold["g_grad_slices"].items():
        assert rel(out["g_grads"][n].flatten()[:64], s) < max(2e-3, 3 * gold["g_grad_slice_noise"][n]), n
    for n, s in gold["d_grad_slices"].items():
        assert rel(out["d_grads"][n].flatten()[:64], s) < max(2e-3, 3 * gold["d_grad_slice_noise"][n]), n
    assert out["g_grads"]["ssl_proj.weight"] is None    # no gradient reaches ssl_proj (models.py:912-921)


def test_mel_filterbank_properties():
    """librosa is absent: the slaney filterbank restatement is checked against its defining properties."""
    import numpy as np
    from oracle.melbank import slaney_mel

    m = slaney_mel(32000, 2048, 128, 0.0, None)
    assert m.shape == (128, 1025) and (m >= 0).all()
    peaks = m.argmax(axis=1)
    assert (np.diff(peaks) > 0).all()                      # centre frequencies increase
    # slaney norm: each triangle has (approximately) unit area in Hz
    hz_per_bin = 16000 / 1024
    area = m.sum(axis=1) * hz_per_bin
    assert np.allclose(area[8:], 1.0, atol=0.12)
    # below 1 kHz the scale is linear: equal spacing of the first filters' peaks
    lin = peaks[: 10]
    assert np.abs(np.diff(lin, 2)).max() <= 1


def test_oracle_s1_matches_reference_fixture():
    import yaml
    from oracle import s1_step as OS
    from util_fill import s1_batch

    gold = torch.load(os.path.join(HERE, "golden", "s1_small.pt"), weights_only=False)
    keys = json.load(open(os.path.join(HERE, "golden", "state_dict_keys.json")))
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "gpt.yaml")))
    sd = {k: v.requires_grad_(True) for k, v in _filled(keys["s1"], 3).items()}
    c = gold["config"]
    b = s1_batch(c["B"], c["x_len"], c["y_len"])
    loss, acc, _ = OS.forward_old(sd, cfg, b["phoneme_ids"], torch.tensor(c["x_lens"]), b["semantic_ids"],
                                  torch.tensor(c["y_lens"]), b["bert_feature"])
    assert abs(float(loss) - gold["loss"]) <= 1e-4 * gold["loss"]
    assert abs(float(acc) - gold["acc"]) < 1e-6
    names = list(gold["grad_slices"])
    grads = torch.autograd.grad(loss, [sd[n] for n in names])
    for n, g in zip(names, grads):
        assert rel(g.flatten()[:96], gold["grad_slices"][n]) < 2e-3, n


def test_oracle_s1_dpo_matches_reference_fixture():
    import yaml
    from oracle import s1_step as OS
    from util_fill import s1_batch

    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "gpt.yaml")))
    keys = json.load(open(os.path.join(HERE, "golden", "state_dict_keys.json")))
    for gold in torch.load(os.path.join(HERE, "golden", "s1_dpo.pt"), weights_only=False)["cases"]:
        c = gold["config"]
        b = s1_batch(c["B"], c["x_len"], c["y_len"])
        y_lens = torch.tensor(c["y_lens"])
        torch.manual_seed(c["seed"])                 # same draws as the reference's make_reject_y
        ry, rl = OS.make_reject_y(b["semantic_ids"], y_lens)
        assert torch.equal(ry, gold["reject_y"]) and torch.equal(rl, gold["reject_y_lens"])
        sd = {k: v.requires_grad_(True) for k, v in _filled(keys["s1"], 3).items()}
        torch.manual_seed(c["seed"])
        loss, acc, (chosen, rejected, loss_2) = OS.forward_dpo(sd, cfg, b["phoneme_ids"], torch.tensor(c["x_lens"]),
                                                               b["semantic_ids"], y_lens, b["bert_feature"])
        assert abs(float(loss) - gold["loss"]) <= 1e-4 * gold["loss"] and abs(float(acc) - gold["acc"]) < 1e-6
        assert torch.allclose(chosen, gold["chosen_logps"], rtol=1e-4) and torch.allclose(rejected, gold["rejected_logps"], rtol=1e-4)
        assert abs(float(loss_2) - gold["loss_dpo"]) <= 2e-3 * gold["loss_dpo"] + 1e-9
        names = list(gold["grad_slices"])
        grads = torch.autograd.grad(loss, [sd[n] for n in names])
        for n, g in zip(names, grads):
            assert rel(g.flatten()[:96], gold["grad_slices"][n]) < 2e-3, (c["seed"], n)


def test_oracle_s1_decoding_matches_reference_fixture():
    """KV-cache decoding: same token sequences as the reference's infer_panel_naive for the same noise table"""
    import yaml
    from oracle import s1_step as OS
    sys.path.insert(0, os.path.join(HERE, "golden"))
    from make_golden_s1_inputs import infer_inputs

    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "gpt.yaml")))
    keys = json.load(open(os.path.join(HERE, "golden", "state_dict_keys.json")))
    sd = _filled(keys["s1"], 3)
    d = infer_inputs()
    with torch.no_grad():
        for gold in torch.load(os.path.join(HERE, "golden", "s1_infer.pt"), weights_only=False)["cases"]:
            a = dict(gold["args"])
            prompt = d["prompts"] if a.pop("prompt") else None
            y, idx, logits = OS.infer_panel_naive(sd, cfg, d["x"], prompt, d["bert"], d["q"], **a)
            assert rel(logits[0][0], gold["logits0"]) < 1e-4 and rel(logits[7][0], gold["logits7"]) < 1e-4
            assert torch.equal(y, gold["y"].long()) and idx == gold["idx"] and len(logits) == gold["steps"]
            assert rel(logits[-1][0], gold["logits_last"]) < 1e-4


def test_oracle_s1_batch_decoding_matches_reference_fixture():
    """the TTS default path (infer_panel_batch_infer): padded batch, rows stopping at different steps, early stop"""
    import yaml
    from oracle import s1_step as OS
    sys.path.insert(0, os.path.join(HERE, "golden"))
    from make_golden_s1_inputs import batch_infer_inputs

    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "gpt.yaml")))
    keys = json.load(open(os.path.join(HERE, "golden", "state_dict_keys.json")))
    sd = _filled(keys["s1"], 3)
    d = batch_infer_inputs()
    with torch.no_grad():
        for gold in torch.load(os.path.join(HERE, "golden", "s1_batch_infer.pt"), weights_only=False)["cases"]:
            a = dict(gold["args"])
            rows = a.pop("rows")
            ys, idxs = OS.infer_panel_batch_infer(sd, cfg, [d["x"][r] for r in rows], d["prompts"][rows],
                                                  [d["bert"][r] for r in rows], d["q"][:, rows], **a)
            assert idxs == gold["idx"]
            for y, g in zip(ys, gold["y"]):
                assert torch.equal(y.long(), g.long())


def test_oracle_scaled_adam_matches_reference_trajectory():
    from oracle import s1_step as OS

    gold = torch.load(os.path.join(HERE, "golden", "s1_small.pt"), weights_only=False)["scaled_adam"]
    params = {k: v.clone() for k, v in gold["init"].items()}
    opt = OS.ScaledAdamRef(params, lr=0.01, clipping_update_period=4)
    for step, (grads, want) in enumerate(zip(gold["grads"], gold["traj"])):
        opt.step(grads)
        opt.lr = 0.002
        for k in params:
            assert torch.allclose(params[k], want[k], rtol=2e-5, atol=1e-6), (step, k)


def test_product_mel_filterbank_is_the_oracle_filterbank_bit_for_bit():
    """the filterbank the product multiplies with (module/mel_processing.py::mel_filterbank) against the oracle's restatement
    of librosa 0.9.2's `filters.mel` (oracle/melbank.py): every float32 identical, at the s2.json configuration and two
    others -- the GPU test of spec_to_mel then checks the matmul against THIS matrix, not against itself"""
    import numpy as np

    from easevoice_trainer_amd.module import mel_processing as PM
    from oracle.melbank import slaney_mel

    for sr, n_fft, n_mels, fmin, fmax in ((32000, 2048, 128, 0.0, None), (22050, 1024, 80, 0.0, 8000.0),
                                          (16000, 512, 40, 50.0, 7600.0)):
        a, b = PM.mel_filterbank(sr, n_fft, n_mels, fmin, fmax), slaney_mel(sr, n_fft, n_mels, fmin, fmax)
        assert a.dtype == np.float32 and a.shape == b.shape == (n_mels, 1 + n_fft // 2)
        assert np.array_equal(a, b), float(np.abs(a - b).max())


def test_mel_filterbank_known_answer_computed_by_hand():
    """sr = 2000, n_fft = 8 (bins 0, 250, 500, 750, 1000 Hz), two filters up to 1000 Hz: everything sits in the LINEAR part
    of the slaney scale (mel = f / (200/3)), so the edges are 0, 1000/3, 2000/3, 1000 Hz, the triangles are
    (0, 333.3, 666.7) and (333.3, 666.7, 1000), and the slaney norm is 2 / 666.67 = 0.003 for both:
        filter 0: bin 250 -> min(250/333.3, 416.7/333.3) = 0.75,  bin 500 -> min(1.5, 166.7/333.3) = 0.5
        filter 1: bin 500 -> min(166.7/333.3, 1.5) = 0.5,         bin 750 -> min(1.25, 250/333.3) = 0.75"""
    import numpy as np

    from easevoice_trainer_amd.module import mel_processing as PM
    from oracle.melbank import slaney_mel

    want = np.array([[0, 0.75 * 0.003, 0.5 * 0.003, 0, 0], [0, 0, 0.5 * 0.003, 0.75 * 0.003, 0]], np.float64)
    for fn in (PM.mel_filterbank, slaney_mel):
        got = fn(2000, 8, 2, 0.0, 1000.0).astype(np.float64)
        assert got.shape == (2, 5) and np.allclose(got, want, rtol=1e-6, atol=1e-9), got
