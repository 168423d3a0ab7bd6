"""Worker of tests/test_host_cpu.py::test_s1_engine_data_parallel_gloo: one rank of a CPU data-parallel run of the s1 engine
(HIP launches emulated, gloo all-reduce), or -- with world size 1 -- the single-process run over the same micro-batches."""
import os
import sys

import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from cpu_emu import cpu_emulation_s1  # noqa: E402
from util_fill import fill_module  # noqa: E402


def small_cfg():
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "gpt.yaml")))
    cfg["model"].update(hidden_dim=64, embedding_dim=64, head=4, n_layer=4, linear_units=256)
    if os.environ.get("EVT_TEST_DPO", "0") == "1":       # the DPO forward: the block stack is walked twice per backward
        cfg.setdefault("train", {})["if_dpo"] = True
    return cfg


def batch(rank, i):
    g = torch.Generator().manual_seed(1000 * rank + i)
    B, x_len, y_len = 2, 6 + i, 9 + rank
    return dict(phoneme_ids=torch.randint(0, 732, (B, x_len), generator=g),
                phoneme_ids_len=torch.tensor([x_len, x_len - 2]),
                semantic_ids=torch.randint(0, 1024, (B, y_len), generator=g),
                semantic_ids_len=torch.tensor([y_len, y_len - 3]),
                bert_feature=torch.randn(B, 1024, x_len, generator=g))


def main():
    out = sys.argv[1]
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    from easevoice_trainer_amd.dist import GradReducer
    from easevoice_trainer_amd.train.s1_engine import S1Engine

    torch.set_num_threads(2)
    with cpu_emulation_s1():
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group("gloo", rank=rank, world_size=world)
        eng = S1Engine(small_cfg(), "cpu", torch.float32, reducer=GradReducer(world) if world > 1 else None)
        fill_module(eng.model, 5)
        eng.model.eval()                      # dropout off: the comparison is about the gradient exchange
        if world > 1:
            eng.reducer.broadcast_params(eng.arena.param)
            stepped = []
            for i in range(5):
                torch.manual_seed(77 + 10 * rank + i)     # make_reject_y draws from the global generator (DPO case)
                stepped.append(eng.micro_step(batch(rank, i), i)[2])
            assert stepped == [False] * 4 + [True]
            # one collective per piece (the test model is far below the bucket size): three with two cuts, one without
            pieces = 1 + len([c for c in os.environ.get("EVT_DP_S1_CUTS", "16,8").split(",") if c.strip() != "" and int(c) < 3])
            if os.environ.get("EVT_TEST_DPO", "0") == "1":
                pieces = 1                    # no range is final before the second traversal's backward: one piece
            assert eng.reducer.stats["all_reduce"] == pieces, (eng.reducer.stats, pieces)
        else:                                 # the same ten micro-batches in one process, mean of the two ranks' sums
            for r in range(2):
                for i in range(5):
                    torch.manual_seed(77 + 10 * r + i)
                    eng.micro_step(batch(r, i), 1)
            eng.arena.grad.mul_(0.5)
            eng.optimizer.step()
        torch.save(eng.arena.param.detach().clone(), out + str(rank))
        if world > 1:
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
