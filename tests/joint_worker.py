"""One rank of tests/test_host_cpu.py::test_joint_launcher_world4: tools/launch_joint.py's run_rank over gloo on the CPU,
world of 4 = s1 on ranks {0, 1} (the real S1Engine on a toy model, HIP launches emulated) + a stand-in job on ranks {2, 3}
that all-reduces inside ITS group every step (the s2 engine has no CPU emulation of its optimiser side).  Checks group
isolation from the inside: a rank's group sum must be the sum over its own group's ranks only."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from cpu_emu import cpu_emulation_s1  # noqa: E402
from dp_worker_s1 import small_cfg  # noqa: E402


class ToyS1:
    unit = "tokens"

    def __init__(self, dev, reducer, rig, args):
        from easevoice_trainer_amd.train.s1_engine import S1Engine
        from util_fill import s1_batch

        self.eng = S1Engine(small_cfg(), dev, torch.float32, reducer=reducer)
        self.batch = s1_batch(2, 6, 10, seed=50 + rig)
        self.idx = 0
        self.units_per_step = 2 * 16

    def params(self):
        return [self.eng.arena.param]

    def step(self):
        self.eng.micro_step(self.batch, self.idx)
        self.idx += 1


class StubS2:
    unit = "audio-s"

    def __init__(self, dev, reducer, rig, args):
        self.reducer, self.rig = reducer, rig
        self.flat = torch.zeros(1000)
        self.units_per_step = 8
        self.world_rank = int(os.environ["RANK"])

    def params(self):
        return [self.flat]

    def step(self):
        g = torch.full((1000,), float(self.world_rank))
        self.reducer.all_reduce(g)
        assert float(g[0]) == 2.0 + 3.0, float(g[0])        # ranks 2 and 3 only: the s1 ranks are not in this sum


def main():
    from tools import launch_joint as LJ

    args = LJ.parser().parse_args(sys.argv[1:])
    with cpu_emulation_s1():
        LJ.run_rank(args, make=lambda role, dev, red, rig, a: (ToyS1 if role == "s1" else StubS2)(dev, red, rig, a))


if __name__ == "__main__":
    main()
