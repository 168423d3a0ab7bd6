"""CPU, build container only: randomised differential tests of the host-side pieces against the REFERENCE's own code,
imported read-only from /root/reference through oracle/refshim.py.  They widen what the committed golden fixtures pin
(one seeded case each) to a few hundred random cases; on a machine without the reference checkout (the GPU box) the
whole module is skipped -- nothing here is needed by the -m gpu tier."""
import os
import random
import sys

import pytest
import torch

from oracle import refshim

if not refshim.reference_available():
    pytest.skip("reference checkout not present", allow_module_level=True)
refshim.install()

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

from easevoice_trainer_amd.auto_reg import utils as PU  # noqa: E402
from easevoice_trainer_amd.train import dataset as D  # noqa: E402


class Stub(torch.utils.data.Dataset):
    def __init__(self, lengths):
        self.lengths = lengths

    def __len__(self):
        return len(self.lengths)

    def get_sample_length(self, i):
        return self.lengths[i]


def test_s2_bucket_sampler_random_cases():
    from src.easevoice.module import data_utils as DU

    rng = random.Random(1)
    for case in range(40):
        n = rng.randint(5, 400)
        lengths = [rng.randint(1, 2300) for _ in range(n)]
        bounds = sorted(rng.sample(range(10, 2200), rng.randint(2, 12)))
        bs, world = rng.randint(1, 9), rng.randint(1, 4)
        inside = [v for v in lengths if bounds[0] < v <= bounds[-1]]
        if not inside:
            continue
        for rank in range(world):
            ref = DU.DistributedBucketSampler(Stub(lengths), bs, list(bounds), num_replicas=world, rank=rank, shuffle=True)
            ours = D.S2BucketSampler(lengths, bs, list(bounds), num_replicas=world, rank=rank)
            for epoch in (0, rng.randint(1, 50)):
                ref.set_epoch(epoch)
                ours.set_epoch(epoch)
                assert list(iter(ours)) == list(iter(ref)), (case, rank, epoch)
            assert len(ours) == len(ref) and ours.boundaries == ref.boundaries


def test_s1_bucket_sampler_random_cases():
    from src.easevoice.soundstorm.auto_reg.data import bucket_sampler as BS

    rng = random.Random(2)
    for case in range(40):
        n = rng.randint(1, 300)
        secs = [round(rng.uniform(0.2, 30.0), 2) for _ in range(n)]
        bs, world = rng.randint(1, 16), rng.randint(1, 4)
        drop = rng.random() < 0.3
        for rank in range(world):
            ref = BS.DistributedBucketSampler(Stub(secs), num_replicas=world, rank=rank, batch_size=bs, drop_last=drop,
                                              seed=case)
            ours = D.S1BucketSampler(Stub(secs), bs, num_replicas=world, rank=rank, drop_last=drop, seed=case)
            for epoch in (0, rng.randint(1, 30)):
                ref.set_epoch(epoch)
                ours.set_epoch(epoch)
                assert list(iter(ours)) == list(iter(ref)), (case, rank, epoch)
            assert len(ours) == len(ref)


def test_s2_collate_random_batches():
    from src.easevoice.module import data_utils as DU

    g = torch.Generator().manual_seed(3)
    rng = random.Random(3)
    for case in range(25):
        items_ref, items = [], []
        for _ in range(rng.randint(1, 7)):
            frames = rng.randint(33, 300)
            ssl_t = frames + rng.choice([0, 0, 1])
            ssl = torch.randn(1, 8, ssl_t, generator=g)
            spec = torch.rand(5, frames, generator=g)
            wav = torch.randn(1, frames * 640 + rng.randint(0, 639), generator=g)
            text = torch.randint(0, 700, (rng.randint(1, 40),), generator=g).float()
            items_ref.append((ssl, spec, wav, text))
            items.append((ssl, wav, text, frames, True))
        ref = DU.TextAudioSpeakerCollate()(items_ref)
        ours, order = D.collate_s2(items, 5)
        for row, src in enumerate(order):
            ours[2][row, :, :items[src][3]] = items_ref[src][1]
        for a, b in zip(ours, ref):
            assert a.dtype == b.dtype and torch.equal(a, b), case


def test_s1_collate_random_batches():
    from src.easevoice.soundstorm.auto_reg.data import dataset as DS

    g = torch.Generator().manual_seed(4)
    rng = random.Random(4)
    ref_ds = object.__new__(DS.Text2SemanticDataset)
    ref_ds.PAD = 1024
    ours = object.__new__(D.S1SemanticTable)
    ours.PAD = 1024
    for case in range(25):
        ex = []
        for i in range(rng.randint(1, 6)):
            n_ph, n_se = rng.randint(1, 50), rng.randint(1, 200)
            ex.append(dict(idx=i, phoneme_ids=[rng.randint(0, 700) for _ in range(n_ph)], phoneme_ids_len=n_ph,
                           semantic_ids=[rng.randint(0, 1023) for _ in range(n_se)], semantic_ids_len=n_se,
                           bert_feature=torch.randn(1024, n_ph, generator=g) if rng.random() < 0.7 else None))
        a, b = ours.collate(ex), ref_ds.collate(ex)
        assert a["ids"] == b["ids"]
        for k in ("phoneme_ids", "phoneme_ids_len", "semantic_ids", "semantic_ids_len", "bert_feature"):
            assert a[k].dtype == b[k].dtype and torch.equal(a[k], b[k]), (case, k)


def test_make_reject_y_random_seeds():
    from src.easevoice.soundstorm.auto_reg.models import utils as RU

    g = torch.Generator().manual_seed(5)
    for seed in range(60):
        B, T = int(torch.randint(1, 6, (1,), generator=g)), int(torch.randint(2, 90, (1,), generator=g))
        y = torch.randint(0, 1025, (B, T), generator=g)
        lens = torch.randint(1, T + 1, (B,), generator=g)
        torch.manual_seed(seed)
        ry, rl = RU.make_reject_y(y, lens)
        after_ref = torch.randint(0, 10 ** 6, (1,)).item()       # the generator ends in the same state
        torch.manual_seed(seed)
        oy, ol = PU.make_reject_y(y, lens)
        after_ours = torch.randint(0, 10 ** 6, (1,)).item()
        assert torch.equal(ry, oy) and torch.equal(rl, ol) and after_ref == after_ours, seed


def test_sampling_restatement_random_logits():
    """the oracle's logits_to_probs (which the GPU sampling kernel is checked against) vs the reference's"""
    from oracle import s1_step as OS
    from src.easevoice.soundstorm.auto_reg.models import utils as RU

    g = torch.Generator().manual_seed(6)
    rng = random.Random(6)
    for case in range(60):
        B, V = rng.randint(1, 4), rng.choice([1024, 1025, 300])
        logits = torch.randn(B, V, generator=g) * rng.choice([0.5, 3.0, 10.0])
        prev = torch.randint(0, V, (B, rng.randint(0, 80)), generator=g)
        kw = dict(temperature=rng.choice([1.0, 0.7, 1.4, 1e-7]), top_k=rng.choice([None, 1, 5, 15, 5000]),
                  top_p=rng.choice([None, 1, 0.95, 0.5, 0.05]), repetition_penalty=rng.choice([1.0, 1.35, 0.8]))
        want = RU.logits_to_probs(logits.clone(), prev if prev.numel() else None, **kw)
        got = OS.logits_to_probs(logits, prev, kw["temperature"], kw["top_k"], kw["top_p"], kw["repetition_penalty"])
        assert torch.equal(got > 0, want > 0) and torch.allclose(got, want, rtol=1e-6, atol=1e-9), (case, kw)


def test_frame_count_formula_random_lengths():
    from src.easevoice.module.mel_processing import spectrogram_torch

    rng = random.Random(7)
    for _ in range(12):
        n = rng.randint(800, 90000)
        spec = spectrogram_torch(torch.randn(1, n), 2048, 32000, 640, 2048, center=False)
        assert spec.shape == (1, 1025, D.spec_frames(n, 2048, 640)), n


def test_dpo_loss_matches_reference_helper():
    from src.easevoice.soundstorm.auto_reg.models import utils as RU

    g = torch.Generator().manual_seed(8)
    for _ in range(20):
        a, r = torch.randn(5, generator=g) * 30, torch.randn(5, generator=g) * 30
        want = RU.dpo_loss(a, r, 0, 0, 0.2, reference_free=True)[0]
        assert torch.allclose(PU.dpo_loss(a, r, 0.2), want, rtol=1e-6, atol=1e-7)


def test_commons_helpers_random():
    from easevoice_trainer_amd.module import commons as PC
    from src.easevoice.module import commons as RC

    g = torch.Generator().manual_seed(9)
    for case in range(20):
        B, T, Cn, seg = int(torch.randint(1, 6, (1,), generator=g)), int(torch.randint(40, 200, (1,), generator=g)), 7, 32
        x = torch.randn(B, Cn, T, generator=g)                        # reference layout [B, C, T]
        lens = torch.randint(seg, T + 1, (B,), generator=g)
        assert torch.equal(PC.sequence_mask(lens, T), RC.sequence_mask(lens, T))
        assert torch.equal(PC.sequence_mask(lens), RC.sequence_mask(lens))
        torch.manual_seed(case)
        want, ids_ref = RC.rand_slice_segments(x, lens, seg)
        torch.manual_seed(case)
        got, ids = PC.rand_slice_segments(x.transpose(1, 2).contiguous(), lens, seg)     # ours is channels-last
        assert torch.equal(ids, ids_ref) and torch.equal(got.transpose(1, 2), want)
        assert torch.equal(PC.slice_segments(x.transpose(1, 2).contiguous(), ids, seg).transpose(1, 2),
                           RC.slice_segments(x, ids, seg))


def test_lr_schedule_trajectory():
    from easevoice_trainer_amd.auto_reg.optim import WarmupCosineLRSchedule as Ours
    from src.easevoice.soundstorm.auto_reg.modules.lr_schedulers import WarmupCosineLRSchedule as Ref

    class Opt:
        def __init__(self):
            self.param_groups = [dict(lr=0.01), dict(lr=0.01)]

    a, b = Opt(), Opt()
    ra = Ref(a, init_lr=1e-5, peak_lr=1e-2, end_lr=1e-4, warmup_steps=20, total_steps=100)
    rb = Ours(b, init_lr=1e-5, peak_lr=1e-2, end_lr=1e-4, warmup_steps=20, total_steps=100)
    for _ in range(150):
        assert ra.step() == rb.step() and ra.get_last_lr() == rb.get_last_lr()
        assert [g["lr"] for g in a.param_groups] == [g["lr"] for g in b.param_groups]
    # the state a Lightning checkpoint keeps under "lr_schedulers": same keys and values, loadable either way
    sa, sb = ra.state_dict(), rb.state_dict()
    assert set(sa) == set(sb) and all(sa[k] == pytest.approx(sb[k]) for k in sa), (sa, sb)
    c = Opt()
    rc = Ours(c, init_lr=1.0, peak_lr=2.0, end_lr=3.0, warmup_steps=1, total_steps=2)
    rc.load_state_dict(sa)
    assert rc._current_step == 150 and rc.warmup_steps == 20 and rc.get_last_lr() == ra.get_last_lr()
    ra.load_state_dict(sb)
    assert ra._current_step == 150


def test_kmeans_codebook_init_matches_reference():
    """a run without a pretrained generator: the codebook initialised from the first batch, same draws / same centres as
    the reference's EuclideanCodebook, for more and for fewer vectors than codes"""
    from easevoice_trainer_amd.module import models as PM
    from src.easevoice.module import core_vq as RV

    g = torch.Generator().manual_seed(10)
    for n, K, dim, iters in ((700, 64, 24, 50), (40, 64, 24, 50), (300, 16, 8, 7)):
        data = torch.randn(n, dim, generator=g) + 2.0 * torch.randint(0, 4, (n, 1), generator=g)
        ref = RV.EuclideanCodebook(dim, K, kmeans_init=True, kmeans_iters=iters)
        ref.eval()
        ours = PM._Codebook(dim, K)
        ours.KMEANS_ITERS = iters
        torch.manual_seed(n)
        _q, ind_ref = ref(data[None])
        after_ref = torch.randint(0, 10 ** 6, (1,)).item()
        torch.manual_seed(n)
        ours.init_embed_(data)
        after_ours = torch.randint(0, 10 ** 6, (1,)).item()
        assert after_ref == after_ours
        assert torch.equal(ours.embed, ref.embed) and torch.equal(ours.embed_avg, ref.embed_avg)
        assert torch.equal(ours.cluster_size, ref.cluster_size) and float(ours.inited) == float(ref.inited) == 1.0
        # (the nearest-code look-up on the initialised codebook is the HIP select kernel: tests/test_frontend_gpu.py)


def _ref_optim_g(net_g, lr, low, betas, eps):
    """the optimiser construction of src/train/sovits.py:285-313 (the trainer module itself needs tensorboard to import)"""
    te_p = list(map(id, net_g.enc_p.text_embedding.parameters()))
    et_p = list(map(id, net_g.enc_p.encoder_text.parameters()))
    mrte_p = list(map(id, net_g.enc_p.mrte.parameters()))
    base = filter(lambda p: id(p) not in te_p + et_p + mrte_p and p.requires_grad, net_g.parameters())
    return torch.optim.AdamW([{"params": base, "lr": lr}, {"params": net_g.enc_p.text_embedding.parameters(), "lr": low},
                              {"params": net_g.enc_p.encoder_text.parameters(), "lr": low},
                              {"params": net_g.enc_p.mrte.parameters(), "lr": low}], lr, betas=betas, eps=eps)


def test_s2_checkpoints_resume_in_both_directions(tmp_path):
    """G_*.pth written by the reference's save_checkpoint resumes here, and ours resumes there: model tensors and every
    parameter's AdamW moments land on the same parameter (the optimiser state is numbered by position, and ssl_proj
    sits in the reference's first group without ever getting state)"""
    import json
    from easevoice_trainer_amd.module import models as PM
    from easevoice_trainer_amd.runtime import FlatAdamW, ParamArena
    from easevoice_trainer_amd.train.s2_engine import S2Engine
    from easevoice_trainer_amd.utils import ckpt as PCK
    from src.easevoice.module import models as RM
    from src.utils.path import ckpt as RCK
    from util_fill import fill_module

    hps = json.load(open(os.path.join(os.path.dirname(HERE), "configs", "s2.json")))
    lr, low, betas, eps = 1e-4, 4e-5, (0.8, 0.99), 1e-9
    ref = RM.SynthesizerTrn(1025, 32, n_speakers=300, **hps["model"])
    fill_module(ref, 1)
    ropt = _ref_optim_g(ref, lr, low, betas, eps)
    g = torch.Generator().manual_seed(0)
    for n, p in ref.named_parameters():
        if not n.startswith("ssl_proj."):
            p.grad = torch.randn(p.shape, generator=g) * 1e-3
    ropt.step()
    for grp in ropt.param_groups:
        grp["initial_lr"] = grp["lr"]              # what ExponentialLR adds (sovits.py:368-376)
    p_ref = str(tmp_path / "G_ref.pth")
    RCK.save_checkpoint(ref, ropt, lr, 3, p_ref)
    ref_state = {n: ropt.state[p] for n, p in ref.named_parameters() if p in ropt.state}
    assert "ssl_proj.weight" not in ref_state and len(ref_state) == len(list(ref.parameters())) - 2

    # ---- reference -> ours ----
    ours = PM.SynthesizerTrn(1025, 32, n_speakers=300, **hps["model"])
    arena = ParamArena(ours, "cpu")
    opt = FlatAdamW(arena, S2Engine.g_param_groups(ours, lr, low), betas=betas, eps=eps)
    _, _, got_lr, it = PCK.load_checkpoint(p_ref, ours, opt)
    assert (got_lr, it) == (lr, 3) and opt.step_count == 1
    named = dict(ours.named_parameters())
    for n, p in ref.named_parameters():
        assert torch.equal(named[n].detach(), p.detach()), n
        b, e = arena.range_of(n, p.numel())
        if n in ref_state:
            assert torch.equal(opt.exp_avg[b:e].view(p.shape), ref_state[n]["exp_avg"]), n
            assert torch.equal(opt.exp_avg_sq[b:e].view(p.shape), ref_state[n]["exp_avg_sq"]), n
        else:
            assert not opt.exp_avg[b:e].any()
    assert [g_["lr"] for g_ in opt.param_groups] == [lr, low, low, low]

    # ---- ours -> reference ----
    p_ours = str(tmp_path / "G_ours.pth")
    PCK.save_checkpoint(ours, opt, lr, 4, p_ours)
    ref2 = RM.SynthesizerTrn(1025, 32, n_speakers=300, **hps["model"])
    ropt2 = _ref_optim_g(ref2, lr, low, betas, eps)
    _, _, got_lr, it = RCK.load_checkpoint(p_ours, ref2, ropt2)
    assert (got_lr, it) == (lr, 4)
    for n, p in ref2.named_parameters():
        assert torch.equal(p.detach(), dict(ref.named_parameters())[n].detach()), n
        if n in ref_state:
            st = ropt2.state[p]
            assert torch.equal(st["exp_avg"], ref_state[n]["exp_avg"]) and float(st["step"]) == 1.0, n
            assert torch.equal(st["exp_avg_sq"], ref_state[n]["exp_avg_sq"]), n
        else:
            assert p not in ropt2.state
    assert [g_["lr"] for g_ in ropt2.param_groups] == [lr, low, low, low]
    # the reference optimiser still steps after the resume (hyper-parameters of the groups are complete)
    for n, p in ref2.named_parameters():
        if not n.startswith("ssl_proj."):
            p.grad = torch.zeros_like(p)
    ropt2.step()


def test_stdout_protocol_lines_equal_reference(capsys):
    """the `<prefix> <json>` lines a trainer process prints are what the reference's own connector prints (and parses)"""
    from easevoice_trainer_amd.utils import connector as PC
    from src.utils.helper import connector as RC
    from src.utils.response import EaseVoiceResponse, ResponseStatus

    ref, ours = RC.MultiProcessOutputConnector(), PC.MultiProcessOutputConnector()
    ref.write_loss(7, 1.25, other={"loss/g/total": 1.25, "learning_rate": 1e-4})
    ref.write_loss(8, 0.5)
    ref.write_log({"epoch": 3, "msg": "x"})
    ref.write_response(EaseVoiceResponse(ResponseStatus.SUCCESS, "Training completed", {"model_path": "/p/m"}))
    ref.write_response(EaseVoiceResponse(ResponseStatus.FAILED, "boom"))
    want = capsys.readouterr().out
    ours.write_loss(7, 1.25, other={"loss/g/total": 1.25, "learning_rate": 1e-4})
    ours.write_loss(8, 0.5)
    ours.write_log({"epoch": 3, "msg": "x"})
    ours.write_response(PC.ResponseStatus.SUCCESS, "Training completed", {"model_path": "/p/m"})
    ours.write_response(PC.ResponseStatus.FAILED, "boom")
    got = capsys.readouterr().out
    assert got == want and want.count("\n") == 5


def test_directory_and_checkpoint_selection_equal_reference(tmp_path):
    from easevoice_trainer_amd.train import helper as PH
    from easevoice_trainer_amd.utils import ckpt as PCK
    from src.train import helper as RH
    from src.utils.path import ckpt as RCK

    for name in ("voice1", "", None):
        a, b = PH.get_sovits_train_dir("/proj", name), RH.get_sovits_train_dir("/proj", name)
        assert os.path.dirname(a) == os.path.dirname(b) and (a == b or not name)
        a, b = PH.get_gpt_train_dir("/proj", name), RH.get_gpt_train_dir("/proj", name)
        assert os.path.dirname(a) == os.path.dirname(b) and (a == b or not name)
    assert PH.train_logs_path == RH.train_logs_path
    rng = random.Random(11)
    for case in range(20):
        d = tmp_path / f"c{case}"
        d.mkdir()
        names = {f"G_{rng.randint(1, 99999)}.pth" for _ in range(rng.randint(1, 6))}
        if rng.random() < 0.4:
            names.add("G_latest.pth")
        names |= {f"D_{rng.randint(1, 99999)}.pth", "notes.txt"}
        for n in names:
            (d / n).write_bytes(b"x")
        assert PCK.latest_checkpoint_path(str(d), "G_*.pth") == RCK.latest_checkpoint_path(str(d), "G_*.pth"), names


def _dataclass_fields_from_source(path, cls):
    """(name, default) pairs of a dataclass read from source text -- the reference's trainer modules cannot be imported
    here (tensorboard / pytorch_lightning are absent)"""
    import ast
    tree = ast.parse(open(path, encoding="utf8").read())
    def lit(v):
        try:
            return ast.literal_eval(v)
        except ValueError:
            return ("name", ast.unparse(v))          # a default that is another module's constant

    for node in ast.walk(tree):
        if isinstance(node, ast.ClassDef) and node.name == cls:
            return [(st.target.id, lit(st.value) if st.value is not None else None)
                    for st in node.body if isinstance(st, ast.AnnAssign)]
    raise KeyError(cls)


def test_params_files_and_configs_equal_reference():
    """the `-c params.json` contract: same fields and defaults as the reference's dataclasses; same training configs"""
    import dataclasses
    import json
    import yaml
    from easevoice_trainer_amd.train.gpt import GPTTrainParams
    from easevoice_trainer_amd.train.sovits import SovitsTrainParams

    root = refshim.REFERENCE_ROOT
    for ours, path, cls in ((SovitsTrainParams, "src/train/sovits.py", "SovitsTrainParams"),
                            (GPTTrainParams, "src/train/gpt.py", "GPTTrainParams")):
        want = _dataclass_fields_from_source(os.path.join(root, path), cls)
        got = [(f.name, f.default) for f in dataclasses.fields(ours)]
        # GPTTrainParams.model_path defaults to the reference's stock checkpoint path there; here "" resolves to the
        # same file through helper.default_pretrained (checked below)
        want = [(n, "" if d == ("name", "gpt_pretrained_model_path") else d) for n, d in want]
        assert got == want, (cls, got, want)
    repo = os.path.dirname(HERE)
    assert json.load(open(os.path.join(repo, "configs", "s2.json"))) == json.load(open(os.path.join(root, "configs", "s2.json")))
    # stock pretrained fallbacks: same files as src/utils/config/__init__.py:34-35 and sovits.py:149-157
    from easevoice_trainer_amd.train import helper as PH
    from src.utils import config as RCFG
    assert os.path.join(RCFG.normalize_root, PH.PRETRAINED_FILES["s1"]) == RCFG.gpt_pretrained_model_path
    assert os.path.join(RCFG.normalize_root, PH.PRETRAINED_FILES["s2G"]) == RCFG.sovits_pretrained_model_path
    assert PH.PRETRAINED_FILES["s2D"] == PH.PRETRAINED_FILES["s2G"].replace("s2G", "s2D")
    ours_cfg = yaml.safe_load(open(os.path.join(repo, "configs", "gpt.yaml")))
    assert ours_cfg == yaml.safe_load(open(os.path.join(root, "configs", "gpt.yaml")))


def test_scaled_adam_random_trajectories():
    """ScaledAdam host logic (batching by shape, scalar branch, size update every 4 steps, gradient clipping from the
    running median with different update periods, lr changes) against the reference's optimiser on random problems"""
    from cpu_emu import cpu_emulation_s1
    from easevoice_trainer_amd.auto_reg.optim import ScaledAdam
    from easevoice_trainer_amd.runtime import ParamArena
    from src.easevoice.soundstorm.auto_reg.modules.optim import ScaledAdam as RefScaledAdam

    rng = random.Random(12)
    g = torch.Generator().manual_seed(12)
    for case in range(6):
        shapes = {}
        for i in range(rng.randint(3, 9)):
            kind = rng.choice(["mat", "mat", "vec", "scalar", "dup"])
            shapes[f"p{i}"] = {"mat": (rng.randint(2, 9), rng.randint(2, 17)), "vec": (rng.randint(2, 20),), "scalar": (1,),
                               "dup": (4, 6)}[kind]
        init = {k: torch.randn(v, generator=g) * (0.5 if len(v) > 1 else 0.2) for k, v in shapes.items()}
        period = rng.choice([3, 4, 7])
        cs = rng.choice([2.0, 1.2])
        ref_p = {k: torch.nn.Parameter(v.clone()) for k, v in init.items()}
        ropt = RefScaledAdam(list(ref_p.values()), lr=0.01, betas=(0.9, 0.95), clipping_scale=cs, clipping_update_period=period,
                             parameters_names=[list(ref_p.keys())], show_dominant_parameters=False)

        class Holder(torch.nn.Module):
            def __init__(self):
                super().__init__()
                for k, v in init.items():
                    setattr(self, k, torch.nn.Parameter(v.clone()))

        with cpu_emulation_s1():
            h = Holder()
            arena = ParamArena(h, "cpu")
            opt = ScaledAdam(arena, lr=0.01, betas=(0.9, 0.95), clipping_scale=cs, clipping_update_period=period)
            params = dict(h.named_parameters())
            for step in range(26):
                scale = 8.0 if rng.random() < 0.15 else 1.0           # spikes so that clipping engages
                arena.zero_grad()
                for k in shapes:
                    gr = torch.randn(shapes[k], generator=g) * 0.1 * scale
                    ref_p[k].grad = gr.clone()
                    params[k].grad.copy_(gr)
                ropt.step()
                opt.step()
                lr = 0.002 if step > 2 else 0.01                      # the scheduler pins the lr after its first steps
                for grp in ropt.param_groups:
                    grp["lr"] = lr
                opt.param_groups[0]["lr"] = lr
                for k in shapes:
                    assert torch.allclose(params[k].detach(), ref_p[k].detach(), rtol=1e-4, atol=3e-6), (case, step, k)


def test_scaled_adam_state_crosses_to_and_from_the_reference_optimizer():
    """src/train/gpt.py:172-177 resumes from a Lightning checkpoint whose "optimizer_states"[0] is the reference
    optimiser's own state_dict (stacked same-shape batches, state under the first parameter of a batch, clipping history in
    the first batch).  ScaledAdam.load_reference_state continues such a run -- same parameters as the reference optimiser
    continuing itself -- and reference_state_dict hands a run back: the reference optimiser loads it and continues like
    ours."""
    from cpu_emu import cpu_emulation_s1
    from easevoice_trainer_amd.auto_reg.optim import ScaledAdam
    from easevoice_trainer_amd.runtime import ParamArena
    from src.easevoice.soundstorm.auto_reg.modules.optim import ScaledAdam as RefScaledAdam

    g = torch.Generator().manual_seed(3)
    shapes = {"emb.w": (7, 6), "a.alpha": (1,), "l0.w": (4, 6), "l0.b": (4,), "b.alpha": (1,), "l1.w": (4, 6), "l1.b": (4,),
              "out.w": (9, 4)}
    init = {k: torch.randn(v, generator=g) * (0.5 if len(v) > 1 else 0.2) for k, v in shapes.items()}
    order = list(shapes)                                  # the reference's parameter order
    kw = dict(lr=0.01, betas=(0.9, 0.95), clipping_scale=2.0, clipping_update_period=5)

    def ref_opt(values):
        ps = {k: torch.nn.Parameter(values[k].clone()) for k in order}
        return ps, RefScaledAdam(list(ps.values()), parameters_names=[order], show_dominant_parameters=False, **kw)

    class Holder(torch.nn.Module):                        # registers its parameters in ANOTHER order than the reference
        def __init__(self, values):
            super().__init__()
            for k in reversed(order):
                self.register_parameter(k.replace(".", "_"), torch.nn.Parameter(values[k].clone()))

    def grads(step):
        gg = torch.Generator().manual_seed(100 + step)
        return {k: torch.randn(shapes[k], generator=gg) * (0.8 if step % 6 == 5 else 0.1) for k in order}

    def ours(values):
        h = Holder(values)
        arena = ParamArena(h, "cpu")
        return h, arena, ScaledAdam(arena, **kw)

    name_of = {k: k.replace(".", "_") for k in order}
    with cpu_emulation_s1():
        # ---- reference runs 13 steps, we continue from its state_dict ----
        rp, ropt = ref_opt(init)
        for step in range(13):
            for k, gr in grads(step).items():
                rp[k].grad = gr
            ropt.step()
        sd = ropt.state_dict()
        h, arena, opt = ours({k: rp[k].detach() for k in order})
        opt.load_reference_state(sd, [name_of[k] for k in order])
        assert opt.step_count == 13
        params = dict(h.named_parameters())
        for step in range(13, 25):
            gs = grads(step)
            arena.zero_grad()
            for k in order:
                rp[k].grad = gs[k]
                params[name_of[k]].grad.copy_(gs[k])
            ropt.step()
            opt.step()
            for k in order:
                assert torch.allclose(params[name_of[k]].detach(), rp[k].detach(), rtol=1e-4, atol=3e-6), (step, k)
        # ---- and back: our state in the reference's layout, loaded by a fresh reference optimiser ----
        back = opt.reference_state_dict([name_of[k] for k in order])
        rp2, ropt2 = ref_opt({k: params[name_of[k]].detach() for k in order})
        ropt2.load_state_dict(back)
        for step in range(25, 36):
            gs = grads(step)
            arena.zero_grad()
            for k in order:
                rp2[k].grad = gs[k]
                params[name_of[k]].grad.copy_(gs[k])
            ropt2.step()
            opt.step()
            for k in order:
                assert torch.allclose(params[name_of[k]].detach(), rp2[k].detach(), rtol=1e-4, atol=3e-6), (step, k)


def test_s1_parameter_order_equals_reference():
    """the s1 resume file stores the optimiser state by parameter INDEX in named_parameters() order (the reference's
    configure_optimizers, t2s_lightning_module.py:94-108): names, order and shapes of our Text2SemanticDecoder equal the
    reference's, so an index means the same tensor on both sides"""
    import yaml
    from easevoice_trainer_amd.auto_reg.t2s_model import Text2SemanticDecoder as Ours
    from src.easevoice.soundstorm.auto_reg.models.t2s_model import Text2SemanticDecoder as Ref

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(repo, "configs", "gpt.yaml")))
    cfg["model"].update(n_layer=2)                       # two layers show the pattern
    ref = Ref(config=cfg, top_k=3)
    ours = Ours(cfg)
    a = [(n, tuple(p.shape)) for n, p in ref.named_parameters()]
    b = [(n, tuple(p.shape)) for n, p in ours.named_parameters()]
    assert a == b
    assert [k for k in ref.state_dict()] == [k for k in ours.state_dict()]
