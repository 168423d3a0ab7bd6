"""CPU tests of the host-side surface: checkpoint layout, stdout protocol, directory naming, resume-file selection,
flat-arena bookkeeping and the data-parallel reducer over gloo (world_size 2)."""
import io
import json
import os
import sys
from contextlib import redirect_stdout

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Opt:
    def __init__(self):
        self.sd = {"state": {0: {"step": torch.tensor(3.0)}}, "param_groups": [{"lr": 1e-4}]}

    def state_dict(self):
        return self.sd

    def load_state_dict(self, sd):
        self.sd = sd


def test_checkpoint_layout_and_latest_selection(tmp_path):
    from easevoice_trainer_amd.utils import ckpt

    m = torch.nn.Linear(4, 3)
    opt = _Opt()
    p1 = tmp_path / "G_100.pth"
    p2 = tmp_path / "G_2333.pth"
    ckpt.save_checkpoint(m, opt, 1e-4, 7, str(p1))
    ckpt.save_checkpoint(m, opt, 1e-4, 9, str(p2))
    d = torch.load(p2, weights_only=False)
    assert set(d) == {"model", "iteration", "optimizer", "learning_rate"} and d["iteration"] == 9
    assert ckpt.latest_checkpoint_path(str(tmp_path), "G_*.pth").endswith("G_2333.pth")
    ckpt.save_checkpoint(m, opt, 1e-4, 11, str(tmp_path / "G_latest.pth"))
    assert ckpt.latest_checkpoint_path(str(tmp_path), "G_*.pth").endswith("G_latest.pth")   # prefers *latest*
    m2 = torch.nn.Linear(4, 3)
    _, _, lr, it = ckpt.load_checkpoint(str(tmp_path / "G_latest.pth"), m2, _Opt())
    assert it == 11 and lr == 1e-4 and torch.equal(m2.weight, m.weight)
    # shape-mismatched keys keep the model's own tensor (ckpt.py:33-46)
    m3 = torch.nn.Linear(5, 3)
    ckpt.load_checkpoint(str(p1), m3, None)
    assert m3.weight.shape == (3, 5)
    with pytest.raises(IndexError):
        ckpt.latest_checkpoint_path(str(tmp_path), "D_*.pth")


def test_stdout_protocol_lines():
    from easevoice_trainer_amd.utils.connector import MultiProcessOutputConnector, ResponseStatus

    buf = io.StringIO()
    with redirect_stdout(buf):
        c = MultiProcessOutputConnector()
        c.write_loss(10, 1.5, {"loss/d/total": 2.0})
        c.write_response(ResponseStatus.SUCCESS, "Finish train sovits", {"model_path": "/x"})
    l1, l2 = buf.getvalue().strip().splitlines()
    assert l1.startswith("loss-of-easevoice ") and json.loads(l1.split(" ", 1)[1]) == {"step": 10, "loss": 1.5, "loss/d/total": 2.0}
    r = json.loads(l2[len("response-of-easevoice "):])
    assert r == {"status": "success", "message": "Finish train sovits", "data": {"model_path": "/x"}, "uuid": None}


def test_dirs_and_resume_file_selection():
    from easevoice_trainer_amd.train.gpt import GPTTrain
    from easevoice_trainer_amd.train.helper import get_gpt_train_dir, get_sovits_train_dir

    assert get_sovits_train_dir("/p", "n") == "/p/models/sovits_train/n"
    assert get_gpt_train_dir("/p", "n") == "/p/models/gpt_train/n"
    assert "/models/sovits_train/sovits_" in get_sovits_train_dir("/p", "")
    assert GPTTrain._get_newest_ckpt(["epoch=1-step=10.ckpt", "epoch=12-step=3.ckpt", "x.txt", "epoch=2-step=99.ckpt"]) == "epoch=12-step=3.ckpt"
    assert GPTTrain._get_newest_ckpt([]) is None


def test_param_arena_views_and_state_dict():
    from easevoice_trainer_amd.runtime import ParamArena

    m = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    before = {k: v.clone() for k, v in m.state_dict().items()}
    a = ParamArena(m, "cpu")
    assert all(torch.equal(m.state_dict()[k], v) for k, v in before.items())
    a.param.zero_()
    assert float(m[0].weight.abs().sum()) == 0.0          # parameters are views of the arena
    m[1].bias.grad.fill_(2.0)
    b, e = a.range_of("1.bias", 3)
    assert torch.equal(a.grad[b:e], torch.full((3,), 2.0)) and b % 64 == 0
    m.load_state_dict(before)                               # load_state_dict copies INTO the views
    assert torch.equal(a.param[a.offsets["0.weight"]: a.offsets["0.weight"] + 35].view(7, 5), before["0.weight"])


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    from easevoice_trainer_amd.dist import GradReducer

    dist.init_process_group("gloo", rank=rank, world_size=world)
    r = GradReducer(world, bucket_bytes=4096)             # several buckets
    torch.manual_seed(0)
    params = torch.randn(5000)
    if rank != 0:
        params += 1.0
    r.broadcast_params(params)
    g = torch.full((5000,), float(rank + 1))
    r.all_reduce(g)
    s = torch.tensor([float(rank)])
    r.all_reduce_scalars(s)
    q.put((rank, float(params.sum()), float(g[0]), float(g[-1]), float(s)))
    dist.destroy_process_group()


def test_grad_reducer_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + os.getpid() % 2000
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert out[0][1] == out[1][1]                 # broadcast made the parameters identical
    for _, _, g0, g1, s in out:
        assert g0 == 3.0 and g1 == 3.0            # 1 + 2 summed over every bucket
        assert s == 0.5                           # averaged scalar


def test_dp_flat_average_equals_full_batch():
    """the DP recipe of S2Engine (sum-all-reduce of the flat grad arena, 1/world folded into the optimiser's grad_scale)
    equals the gradient of the mean loss over the global batch, for a loss that is a mean over samples"""
    torch.manual_seed(1)
    w = torch.randn(6, requires_grad=True)
    x = torch.randn(8, 6)
    full = ((x @ w) ** 2).mean()
    (gf,) = torch.autograd.grad(full, w)
    shards = [((x[i::2] @ w) ** 2).mean() for i in range(2)]
    gs = sum(torch.autograd.grad(s, w)[0] for s in shards) * 0.5
    assert torch.allclose(gf, gs, atol=1e-6)


def test_stock_pretrained_fallback(tmp_path, monkeypatch):
    """an empty pretrained path in the params file means the stock checkpoints, looked up like the reference does"""
    from easevoice_trainer_amd.train import helper
    from easevoice_trainer_amd.train.gpt import GPTTrain, GPTTrainParams
    from easevoice_trainer_amd.train.sovits import SovitsTrain, SovitsTrainParams

    monkeypatch.chdir(tmp_path)
    monkeypatch.delenv("EVT_PRETRAINED_DIR", raising=False)
    if not os.path.isdir(os.path.join(helper.repo_root(), "models", "pretrained")):
        assert helper.default_pretrained("s2G") == ""
    root = tmp_path / "pre"
    for k, rel in helper.PRETRAINED_FILES.items():
        f = root / rel
        f.parent.mkdir(parents=True, exist_ok=True)
        f.write_bytes(b"x")
    monkeypatch.setenv("EVT_PRETRAINED_DIR", str(root))
    s = SovitsTrain(SovitsTrainParams(project_dir=str(tmp_path), output_model_name="a"))
    assert s.hps["train"]["pretrained_s2G"] == str(root / helper.PRETRAINED_FILES["s2G"])
    assert s.hps["train"]["pretrained_s2D"] == str(root / helper.PRETRAINED_FILES["s2D"])
    s = SovitsTrain(SovitsTrainParams(project_dir=str(tmp_path), output_model_name="b", pretrained_s2G="/x/G.pth",
                                      pretrained_s2D="pretrained/gsv-v2final-pretrained/s2D2333k.pth"))
    assert s.hps["train"]["pretrained_s2G"] == "/x/G.pth"
    assert s.hps["train"]["pretrained_s2D"] == str(root / helper.PRETRAINED_FILES["s2D"])
    g = GPTTrain(GPTTrainParams(project_dir=str(tmp_path), output_model_name="c"))
    assert g.config["pretrained_s1"] == str(root / helper.PRETRAINED_FILES["s1"])
    g = GPTTrain(GPTTrainParams(project_dir=str(tmp_path), output_model_name="d", model_path="/y/s1.ckpt"))
    assert g.config["pretrained_s1"] == "/y/s1.ckpt"
    # ./models/pretrained of the working directory (the service starts the trainer inside its checkout)
    monkeypatch.delenv("EVT_PRETRAINED_DIR")
    cw = tmp_path / "models" / "pretrained" / helper.PRETRAINED_FILES["s1"]
    cw.parent.mkdir(parents=True)
    cw.write_bytes(b"x")
    assert helper.default_pretrained("s1") == str(cw)


def test_spawn_ranks_runs_one_process_per_gpu_and_stops_on_failure(tmp_path):
    """the launcher behind `gpu_ids = "0-1"`: env per rank, a gloo all-reduce across the children, and termination of
    the surviving ranks when one fails"""
    import sys
    import time
    from easevoice_trainer_amd.dist import parse_gpu_ids, spawn_ranks

    assert parse_gpu_ids("0-2-5") == [0, 2, 5] and parse_gpu_ids("3") == [3] and parse_gpu_ids("0,1") == [0, 1]
    ok = tmp_path / "ok.py"
    ok.write_text("""
import os, sys, torch, torch.distributed as dist
dist.init_process_group('gloo', rank=int(os.environ['RANK']), world_size=int(os.environ['WORLD_SIZE']))
t = torch.tensor([float(os.environ['LOCAL_RANK'])])
dist.all_reduce(t)
with open(sys.argv[1] + os.environ['RANK'], 'w') as f:
    f.write(os.environ['LOCAL_RANK'] + ' ' + os.environ['MASTER_ADDR'] + ' ' + str(t.item()))
dist.destroy_process_group()
""")
    codes = spawn_ranks([sys.executable, str(ok), str(tmp_path / "out")], [0, 2])
    assert codes == [0, 0]
    assert (tmp_path / "out0").read_text() == "0 127.0.0.1 2.0" and (tmp_path / "out1").read_text() == "2 127.0.0.1 2.0"
    bad = tmp_path / "bad.py"
    bad.write_text("""
import os, sys, time
if os.environ['RANK'] == '1':
    sys.exit(3)
time.sleep(60)
""")
    t0 = time.time()
    codes = spawn_ranks([sys.executable, str(bad)], [0, 1])
    assert codes[1] == 3 and codes[0] not in (None, 0) and time.time() - t0 < 30


def test_s1_engine_data_parallel_gloo(tmp_path):
    """two ranks of the s1 engine (emulated launches, gloo): gradients accumulated locally over the accumulation window and
    exchanged ONCE at the optimiser step give the parameters of a single process that saw both ranks' micro-batches
    (the reference all-reduces on every micro-batch; the sum is the same)"""
    import sys
    from easevoice_trainer_amd.dist import spawn_ranks

    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dp_worker_s1.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    import subprocess
    subprocess.run([sys.executable, worker, str(tmp_path / "single")], check=True, env=env)
    one = torch.load(tmp_path / "single0")
    # the reduction in three pieces that start inside the last micro-batch's backward (cuts at blocks 2 and 1 of the
    # 4-block test model), and as one piece after it: both = the single process
    for tag, cuts in (("cut", "2,1"), ("plain", "")):
        os.environ["EVT_DP_S1_CUTS"] = cuts
        try:
            codes = spawn_ranks([sys.executable, worker, str(tmp_path / tag)], [0, 1])
        finally:
            del os.environ["EVT_DP_S1_CUTS"]
        assert codes == [0, 0]
        a, b = torch.load(tmp_path / (tag + "0")), torch.load(tmp_path / (tag + "1"))
        assert torch.equal(a, b)                                   # replicas stay identical
        assert torch.allclose(a, one, rtol=1e-5, atol=1e-7) and not torch.equal(a, torch.zeros_like(a)), tag


def test_s1_engine_data_parallel_gloo_dpo(tmp_path):
    """the same equality with train.if_dpo (the forward walks the block stack twice, chosen and rejected targets): with
    cut points configured the reduction must still be ONE piece after the backward -- a hook-driven early piece would
    hand a range to the collective before the second traversal has added its gradients"""
    import subprocess
    import sys
    from easevoice_trainer_amd.dist import spawn_ranks

    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dp_worker_s1.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["EVT_TEST_DPO"] = "1"
    subprocess.run([sys.executable, worker, str(tmp_path / "single")], check=True, env=env)
    one = torch.load(tmp_path / "single0")
    os.environ["EVT_DP_S1_CUTS"] = "2,1"
    os.environ["EVT_TEST_DPO"] = "1"
    try:
        codes = spawn_ranks([sys.executable, worker, str(tmp_path / "dpo")], [0, 1])
    finally:
        del os.environ["EVT_DP_S1_CUTS"], os.environ["EVT_TEST_DPO"]
    assert codes == [0, 0]
    a, b = torch.load(tmp_path / "dpo0"), torch.load(tmp_path / "dpo1")
    assert torch.equal(a, b)
    assert torch.allclose(a, one, rtol=1e-5, atol=1e-7) and not torch.equal(a, torch.zeros_like(a))


def test_split_subgroups_gloo(tmp_path):
    """BASELINE config 5's layout: one world, two sub-communicators (s1 ranks / s2 ranks), each reducing only its own
    gradients through GradReducer(group=...)"""
    import sys
    from easevoice_trainer_amd.dist import spawn_ranks

    w = tmp_path / "sub.py"
    w.write_text("""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from easevoice_trainer_amd.dist import GradReducer, split_subgroups
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', rank=rank, world_size=world)
ga, gb = split_subgroups([0], [1, 2])
mine, n = (ga, 1) if rank == 0 else (gb, 2)
red = GradReducer(n, bucket_bytes=64, group=mine)
g = torch.full((50,), float(rank + 1))
red.all_reduce(g, average=True)
with open(sys.argv[1] + str(rank), 'w') as f:
    f.write(str(g[0].item()) + ' ' + str(g[-1].item()))
dist.destroy_process_group()
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    codes = spawn_ranks([sys.executable, str(w), str(tmp_path / "o")], [0, 1, 2])
    assert codes == [0, 0, 0]
    assert (tmp_path / "o0").read_text() == "1.0 1.0"            # a group of one is left alone
    assert (tmp_path / "o1").read_text() == "2.5 2.5" and (tmp_path / "o2").read_text() == "2.5 2.5"


def test_tensorboard_event_files_decode_with_protobuf(tmp_path):
    """the hand-encoded event records parse with the protobuf runtime against the Event/Summary schema, the TFRecord
    framing carries the masked CRC-32C of length and payload, and CRC-32C matches its published check value"""
    import struct
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    from easevoice_trainer_amd.utils import tb

    assert tb.crc32c(b"123456789") == 0xE3069283
    fd = descriptor_pb2.FileDescriptorProto(name="evt_event.proto", package="evt", syntax="proto3")
    T = descriptor_pb2.FieldDescriptorProto
    val = fd.message_type.add(name="Value")
    val.field.add(name="tag", number=1, type=T.TYPE_STRING, label=T.LABEL_OPTIONAL)
    val.field.add(name="simple_value", number=2, type=T.TYPE_FLOAT, label=T.LABEL_OPTIONAL)
    summ = fd.message_type.add(name="Summary")
    summ.field.add(name="value", number=1, type=T.TYPE_MESSAGE, type_name=".evt.Value", label=T.LABEL_REPEATED)
    ev = fd.message_type.add(name="Event")
    ev.field.add(name="wall_time", number=1, type=T.TYPE_DOUBLE, label=T.LABEL_OPTIONAL)
    ev.field.add(name="step", number=2, type=T.TYPE_INT64, label=T.LABEL_OPTIONAL)
    ev.field.add(name="file_version", number=3, type=T.TYPE_STRING, label=T.LABEL_OPTIONAL)
    ev.field.add(name="summary", number=5, type=T.TYPE_MESSAGE, type_name=".evt.Summary", label=T.LABEL_OPTIONAL)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    Event = message_factory.GetMessageClass(pool.FindMessageTypeByName("evt.Event"))

    w = tb.ScalarWriter(str(tmp_path / "run"))
    w.add_scalars(5, {"loss/g/total": 1.25, "learning_rate": 1e-4})
    w.add_scalars(300000, {"loss/d/total": -2.5})
    w.close()
    blob = open(w.path, "rb").read()
    assert os.path.basename(w.path).startswith("events.out.tfevents.")
    events, pos = [], 0
    while pos < len(blob):
        (n,) = struct.unpack_from("<Q", blob, pos)
        assert struct.unpack_from("<I", blob, pos + 8)[0] == tb._masked(blob[pos:pos + 8])
        rec = blob[pos + 12:pos + 12 + n]
        assert struct.unpack_from("<I", blob, pos + 12 + n)[0] == tb._masked(rec)
        e = Event()
        e.ParseFromString(rec)
        events.append(e)
        pos += 16 + n
    assert [e.step for e in events] == [0, 5, 300000] and events[0].file_version == "brain.Event:2"
    assert [(v.tag, round(v.simple_value, 6)) for v in events[1].summary.value] == [("loss/g/total", 1.25), ("learning_rate", 1e-4)]
    assert events[2].summary.value[0].tag == "loss/d/total" and events[2].summary.value[0].simple_value == -2.5
    assert all(e.wall_time > 1.6e9 for e in events)


def test_single_gpu_id_selects_that_device(monkeypatch):
    """gpu_ids="2" without a launcher trains on card 2 (the reference exports CUDA_VISIBLE_DEVICES=gpu_ids,
    src/train/sovits.py:168); under a launcher LOCAL_RANK wins"""
    from easevoice_trainer_amd.dist import init_process_group_from_env

    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert init_process_group_from_env(gpu_ids="2") == (1, 0, 2)
    assert init_process_group_from_env(gpu_ids="1-3") == (1, 0, 1)
    assert init_process_group_from_env() == (1, 0, 0)
    monkeypatch.setenv("LOCAL_RANK", "5")
    assert init_process_group_from_env(gpu_ids="2") == (1, 0, 5)


def test_param_arena_adjacent_groups_and_prefix_ranges():
    """ParamArena honours a sub-module's arena_adjacent() (the q / k / v projection weights of an attention layer laid out
    back to back, so that they form one dense matrix: hip/conv.py::PackedConv) without touching names, shapes or state_dict
    order; range_of_prefix gives the contiguous range of a sub-model (what the data-parallel step reduces on its own)."""
    from easevoice_trainer_amd.runtime import ParamArena

    class Att(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv_q, self.conv_k, self.conv_v = nn.Linear(64, 64), nn.Linear(64, 64), nn.Linear(64, 64)
            self.out = nn.Linear(64, 8)

        def arena_adjacent(self):
            return [["conv_q.weight", "conv_k.weight", "conv_v.weight"], ["conv_q.bias", "conv_k.bias", "conv_v.bias"]]

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.pre = nn.Linear(8, 64)
            self.att = Att()
            self.post = nn.Linear(8, 3)

    torch.manual_seed(0)
    m = Net()
    keys = list(m.state_dict().keys())
    before = {k: v.clone() for k, v in m.state_dict().items()}
    a = ParamArena(m, "cpu")
    assert list(m.state_dict().keys()) == keys and all(torch.equal(m.state_dict()[k], before[k]) for k in keys)
    o = a.offsets
    w = 64 * 64
    assert o["att.conv_k.weight"] == o["att.conv_q.weight"] + w and o["att.conv_v.weight"] == o["att.conv_q.weight"] + 2 * w
    assert o["att.conv_k.bias"] == o["att.conv_q.bias"] + 64 and o["att.conv_v.bias"] == o["att.conv_q.bias"] + 128
    packed = a.param[o["att.conv_q.weight"]: o["att.conv_q.weight"] + 3 * w].view(192, 64)
    assert torch.equal(packed, torch.cat([m.att.conv_q.weight, m.att.conv_k.weight, m.att.conv_v.weight]))
    lo, hi = a.range_of_prefix("att.")
    names = [n for n in a.names if n.startswith("att.")]
    assert lo == min(o[n] for n in names) and hi == o["post.weight"]       # the sub-model is one contiguous range
    lo2, hi2 = a.range_of_prefix("att.", stop_before="att.out.")
    assert lo2 == lo and hi2 == o["att.out.weight"]
    # gradients are views of the twin arena at the same offsets
    assert m.att.conv_k.weight.grad.data_ptr() == a.grad.data_ptr() + 4 * o["att.conv_k.weight"]


def test_weight_bank_row_ranges_cpu():
    """WeightBank.rows_of: the row range of a contiguous run of convolutions (one sub-discriminator / the vocoder) in the
    multi-tensor gradient launch; non-contiguous selections are refused"""
    from easevoice_trainer_amd.hip import conv as HC

    m = nn.ModuleList([HC.EvtConv1d(8, 16, 3, padding=1, weight_norm=True), HC.EvtConv1d(16, 16, 1),
                       HC.EvtConv1d(16, 32, 5, padding=2, weight_norm=True)])
    bank = HC.WeightBank(m, torch.float32, "cpu")
    bank.build_tables()
    assert bank.rows_of([m[0]]) == (0, 16) and bank.rows_of([m[1], m[2]]) == (16, 64) and bank.rows_of(list(m)) == (0, 64)
    with pytest.raises(HC.L.EvtError):
        bank.rows_of([m[0], m[2]])


def test_s2_data_parallel_program_plumbing_cpu():
    """The data-parallel s2 step's bookkeeping without launching anything (train/s2_engine.py::_program): nine pieces by
    default (D forward, six per-sub-discriminator backward pieces, the generator's backward as ONE piece since round 5 --
    an early generator piece costs more on the GPU than the exchange it hides --, the G optimiser), eleven with
    EVT_DP_G_PIECES=3 (through D and the vocoder | flow + posterior encoder | prior + style encoder); the
    sub-discriminators' arena ranges tile the discriminator arena and their weight-gradient row ranges tile the bank's row
    table; the vocoder's range and rows are contiguous, exclude its conditioning layer's parameters and leave two rest
    ranges; the generator is switched to the cut backward.  (The arithmetic of the overlapped step is checked on the GPU:
    tests/test_zz_dp_overlap_gpu.py.)"""
    import json
    from easevoice_trainer_amd.train.s2_engine import S2Engine

    class FakeReducer:
        world = 2
        active = True

    hps = json.load(open(os.path.join(ROOT, "configs", "s2.json")))
    eng = S2Engine(hps, "cpu", torch.float32, reducer=FakeReducer())
    assert eng.overlap and not eng.net_g.split_backward          # one generator piece: its autograd graph has no cuts
    prog = eng._program()
    assert len(prog) == 9 and prog[0][1] is None and prog[-1][1] is None and all(a is not None for _, a in prog[1:-1])
    assert [n for n, _ in eng.exchange_ranges()] == [f"D piece {i}/6" for i in range(1, 7)] + ["G rest"]
    for ng, n in ((3, 11), (2, 10), (1, 9)):
        eng.dp_g_pieces = ng
        assert len(eng._program()) == n
    eng.dp_d_pieces = 2
    assert len(eng._program()) == 5
    eng.dp_d_pieces = 6
    os.environ["EVT_DP_G_PIECES"] = "3"
    try:
        e3 = S2Engine(hps, "cpu", torch.float32, reducer=FakeReducer())
    finally:
        del os.environ["EVT_DP_G_PIECES"]
    assert e3.net_g.split_backward and len(e3._program()) == 11
    # discriminator: ranges tile the arena in order, row ranges tile the row table
    d = eng.rt_d.arena
    at = 0
    for lo, hi in eng._d_ranges:
        assert lo == at and hi > lo
        at = hi
    assert at == d.numel
    rows = [eng.rt_d.bank.rows_of(c) for c in eng._d_convs]
    assert rows[0][0] == 0 and all(rows[i][1] == rows[i + 1][0] for i in range(5)) and rows[-1][1] == eng.rt_d.bank._nrows
    # generator: the vocoder's parameters (without its conditioning layer) are one range strictly inside the arena
    g = eng.rt_g.arena
    lo, hi = eng._dec_range
    names = [n for n in g.names if lo <= g.offsets[n] < hi]
    assert names and all(n.startswith("dec.") and not n.startswith("dec.cond.") for n in names)
    assert sum(1 for n in g.names if n.startswith("dec.") and not n.startswith("dec.cond.")) == len(names)
    assert 0 < lo < hi < g.numel
    rlo, rhi = eng.rt_g.bank.rows_of(eng._dec_convs)
    assert 0 < rlo < rhi <= eng.rt_g.bank._nrows
    # flow + posterior encoder: one range holding exactly their parameters, disjoint from the vocoder's, one contiguous
    # run of weight-gradient rows; what is left for the last reduction are the prior / style encoders and dec.cond
    flo, fhi = eng._fq_range
    fq = [n for n in g.names if flo <= g.offsets[n] < fhi]
    assert fq and all(n.startswith(("flow.", "enc_q.")) for n in fq)
    assert len(fq) == sum(1 for n in g.names if n.startswith(("flow.", "enc_q.")))
    assert fhi <= lo or flo >= hi
    qlo, qhi = eng.rt_g.bank.rows_of(eng._fq_convs)
    assert qhi > qlo and (qhi <= rlo or qlo >= rhi)
    rest = [n for n in g.names if not (lo <= g.offsets[n] < hi or flo <= g.offsets[n] < fhi)]
    assert rest and all(n.startswith(("enc_p.", "ref_enc.", "dec.cond.", "ssl_proj.", "quantizer.")) for n in rest), rest[:5]
    # the attention layers' projections are adjacent in the generator's arena (what the packed projection needs)
    o = g.offsets
    pre = "enc_p.encoder_ssl.attn_layers.0."
    assert o[pre + "conv_k.weight"] == o[pre + "conv_q.weight"] + 192 * 192 == o[pre + "conv_v.weight"] - 192 * 192


def test_spawned_ranks_share_the_run_name(tmp_path):
    """a run without output_model_name gets a timestamp name; the ranks of one launch must agree on it"""
    import sys
    from easevoice_trainer_amd.dist import spawn_ranks

    w = tmp_path / "w.py"
    w.write_text(f"""
import os, sys, time
sys.path.insert(0, {ROOT!r})
time.sleep(1.2 * int(os.environ['RANK']))          # ranks reach the naming code on different seconds
from easevoice_trainer_amd.train.helper import get_sovits_train_dir
open(os.path.join({str(tmp_path)!r}, 'name' + os.environ['RANK']), 'w').write(get_sovits_train_dir('/p', ''))
""")
    assert spawn_ranks([sys.executable, str(w)], [0, 1]) == [0, 0]
    a, b = (tmp_path / "name0").read_text(), (tmp_path / "name1").read_text()
    assert a == b and "sovits_" in a


def test_attention_dropout_hash_statistics():
    """the counter hash behind the s1 attention dropout mask (csrc/attention.hip: one 32-bit hash decides the keys k and
    k ^ 16 of a query through its two 16-bit halves; python mirror in tests/test_s1_gpu.py, which the GPU parity test
    checks the kernels against): drop rate = p, no structure between neighbouring keys, the two halves of one hash,
    neighbouring queries or neighbouring (batch, head) slices"""
    from test_s1_gpu import _hash_keep

    L_, p = 512, 0.1
    thr16 = int(p * 65536.0 + 0.5)
    drop = torch.stack([(~_hash_keep(seed, bh, L_, thr16)).double() for bh in range(0, 64, 7) for seed in (0, 12345, 0xDEADBEEF)])
    assert abs(drop.mean().item() - p) < 1e-3
    assert abs(drop.mean(dim=2).std().item() - (p * (1 - p) / L_) ** 0.5) < 3e-3        # per-query rates spread like Bernoulli draws
    c = drop - drop.mean()
    var = (c * c).mean()

    def corr(a, b):
        return abs(((a * b).mean() / var).item())

    assert corr(c[:, :, :-1], c[:, :, 1:]) < 0.02 and corr(c[:, :, :-16], c[:, :, 16:]) < 0.02     # key neighbours, hash halves
    assert corr(c[:, :-1, :], c[:, 1:, :]) < 0.02 and corr(c[:-1], c[1:]) < 0.02                   # queries, slices


def test_bench_multi_rank_dry_run(tmp_path):
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one process per rank), on the CPU: gloo
    transport, toy s1 model with emulated launches (tests/bench_dryrun_worker.py).  The path from rendezvous to the
    single JSON line is the one the 8-GPU run takes."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EVT_BENCH_BACKEND="gloo", EVT_BENCH_TINY="1", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29571", os.path.join(root, "tests", "bench_dryrun_worker.py"), "--gpus", "2", "--steps", "5",
           "--warmup", "1", "--workload", "s1", "--s1-batch", "2", "--dtype", "f32", "--no-extras"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["parallelism"].startswith("dp2, reduce-scatter + all-gather for buckets >= 8 MiB from 3 ranks up") and d["config"]["global_batch"] == 4 and d["value"] > 0
    # the line explains its own gradient exchange: one all-reduce of the (toy) arena every fourth micro-step
    assert d["comm"]["all_reduce_per_step"] > 0 and d["comm"]["mib_per_step"] > 0 and d["comm"]["rs_ag_per_step"] == 0
    # a launch with the wrong world size is refused, not silently run as N independent jobs
    cmd1 = [sys.executable, os.path.join(root, "tests", "bench_dryrun_worker.py"), "--gpus", "2", "--workload", "s1"]
    r1 = subprocess.run(cmd1, capture_output=True, text=True, timeout=120, cwd=root,
                        env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r1.returncode != 0 and "WORLD_SIZE=1" in (r1.stderr + r1.stdout)


def test_joint_launcher_world4():
    """BASELINE configs[4] (s1 on some ranks, s2 on the others, one bootstrap, two sub-communicators): the launcher's rank
    logic on the CPU with a world of 4 -- group isolation is asserted inside the workers (tests/joint_worker.py)"""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", "29573", os.path.join(root, "tests", "joint_worker.py"), "--s1-gpus", "0-1", "--s2-gpus", "2-3",
           "--minutes", "0.05", "--warmup", "1", "--check-every", "2", "--backend", "gloo"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["s1"]["n_gpus"] == 2 and d["s2"]["n_gpus"] == 2
    assert d["s1"]["tokens_per_sec"] > 0 and d["s2"]["audio_seconds_per_sec"] > 0
    assert d["s1"]["seconds"] >= 3.0 and d["s2"]["seconds"] >= 3.0          # both groups ran to the deadline


def test_reduce_scatter_all_gather_equals_all_reduce():
    """EVT_DP_RSAG: GradReducer's reduce-scatter + all-gather buckets give the sums its all-reduce buckets give (world 3:
    shards that do not divide the buffer, tails shorter than the world)"""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
           "--master-port", "29577", os.path.join(root, "tests", "rsag_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root,
                       env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
    assert r.returncode == 0, r.stderr[-3000:]


def test_slab_count_policy():
    """slabs per gradient image (hip/conv.py::slab_count): bounded by the memory budget, by the fold cap for ordinary
    layers, lifted for the layers whose parallelism is all position split"""
    from easevoice_trainer_amd.hip.conv import slab_count

    assert slab_count(1024 * 1024 * 5 * 4, 1024) == 3            # 21 MB image: budget-bound
    assert slab_count(128 * 128 * 11 * 4, 128) == 32             # vocoder C = 128: the cap
    assert slab_count(64 * 64 * 11 * 4, 64) == 128               # at most 64 output channels: one or two output tiles
    assert slab_count(16 * 16 * 12 * 4, 16) == 256               # a few KB: as many slabs as the kernel has splits
    assert slab_count(32 * 32 * 11 * 4, 32) == 256
    assert slab_count(0, 1) == 256 and slab_count(1 << 30, 512) == 1
    assert slab_count(128 * 128 * 11 * 4, 128, cap=8) == 8 and slab_count(4 << 20, 512, budget=8 << 20) == 2


def test_compute_dtype_follows_fp16_run(monkeypatch):
    """train.fp16_run (configs/s2.json: true) selects the reference's float16 + GradScaler mode; bfloat16 only on request"""
    import json

    from easevoice_trainer_amd.train.sovits import compute_dtype

    hps = json.load(open(os.path.join(ROOT, "configs", "s2.json")))
    monkeypatch.delenv("EVT_HALF", raising=False)
    assert hps["train"]["fp16_run"] is True and compute_dtype(hps) == torch.float16
    monkeypatch.setenv("EVT_HALF", "bf16")
    assert compute_dtype(hps) == torch.bfloat16
    monkeypatch.setenv("EVT_HALF", "fp8")
    with pytest.raises(ValueError):
        compute_dtype(hps)
    hps["train"]["fp16_run"] = False
    assert compute_dtype(hps) == torch.float32


def test_s1_compute_dtype_follows_precision(monkeypatch):
    """train.precision (configs/gpt.yaml: 16-mixed) selects float16 + loss scaling; bfloat16 on request or by config"""
    import yaml

    from easevoice_trainer_amd.train.gpt import compute_dtype

    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "gpt.yaml")))
    monkeypatch.delenv("EVT_HALF", raising=False)
    assert cfg["train"]["precision"] == "16-mixed" and compute_dtype(cfg) == torch.float16
    monkeypatch.setenv("EVT_HALF", "bf16")
    assert compute_dtype(cfg) == torch.bfloat16
    cfg["train"]["precision"] = "bf16-mixed"
    assert compute_dtype(cfg) == torch.bfloat16
    cfg["train"]["precision"] = "32"
    assert compute_dtype(cfg) == torch.float32


def test_reducer_plan_and_description():
    """GradReducer.plan / describe(ranges): which collective a flat range becomes, per bucket, without any process group"""
    from easevoice_trainer_amd.dist import GradReducer

    mib = 1 << 18                                     # fp32 elements per MiB
    r8 = GradReducer(8, bucket_bytes=64 << 20, rsag="auto")
    assert r8.active and r8.plan(31 * mib) == [(31 << 20, "reduce-scatter + all-gather")]
    assert r8.plan(4 * mib) == [(4 << 20, "all-reduce")]                       # below rsag_min_bytes
    assert [k for _b, k in r8.plan(150 * mib)] == ["reduce-scatter + all-gather"] * 3 and sum(b for b, _k in r8.plan(150 * mib)) == 150 << 20
    r2 = GradReducer(2, rsag="auto")
    assert r2.plan(100 * mib) == [(64 << 20, "all-reduce"), (36 << 20, "all-reduce")]      # two ranks: one link either way
    d = r8.describe([("D piece 1/6", 31 * mib), ("G rest", 4 * mib)])
    assert d.startswith("dp8, reduce-scatter + all-gather for buckets >= 8 MiB from 3 ranks up (after a start-up equality "
                        "check against all-reduce on the group), all-reduce otherwise, buckets of 64 MiB")
    assert "D piece 1/6 31.0 MiB = 1 x reduce-scatter + all-gather" in d and "G rest 4.0 MiB = 1 x all-reduce" in d
    r8.rsag_verified = False                     # what a failed start-up check leaves behind: all-reduce everywhere
    assert r8.plan(31 * mib) == [(31 << 20, "all-reduce")]
    one = GradReducer(1)
    assert not one.active and GradReducer(1, force=True).active
    with pytest.raises(ValueError):
        GradReducer(2, rsag="sometimes")
