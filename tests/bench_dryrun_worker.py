"""One rank of the CPU dry run of `bench.py --gpus N` (tests/test_host_cpu.py::test_bench_multi_rank_dry_run): the bench's
own main() -- rendezvous, parameter broadcast, timed micro-steps with the gradient all-reduce, max-over-ranks timing, the
JSON line -- over gloo, on a toy s1 model whose HIP launches are emulated (tests/cpu_emu.py).  What it cannot cover (the
kernels, RCCL itself) is covered on the GPU box by tests/test_zz_bench_dp_gpu.py."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from cpu_emu import cpu_emulation_s1  # noqa: E402


def main():
    import bench

    with cpu_emulation_s1():
        bench.main()


if __name__ == "__main__":
    main()
