"""MI355X-native GPT-SoVITS training hot path.

HIP runtime switch that must be in the environment BEFORE libamdhip64 is loaded (i.e. before the first `import torch`):

DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 -- ROCm 7.2's graph executor pre-records the AQL packets of kernel nodes ("packet
capture").  In graphs of a few thousand nodes that optimisation loses the ordering between a memset node and the kernel
node behind it on replays after the first: ATen's multi-block reductions (which zero their semaphores with a
hipMemsetAsync node) then return stale results -- the cause of round 1's NaN losses under graph replay of the s2 step.
Measured with a torch-only reproducer (tools/repro_graph_packet_capture.py, 600 reductions in one graph: 134 wrong results per
replay with the default, 0 with the switch off).  `hip_graphs_safe()` tells the engines whether the switch took effect;
they refuse to capture otherwise and keep launching eagerly.
"""
import os
import sys

_FLAG = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"
_torch_loaded_first = "torch" in sys.modules and os.environ.get(_FLAG) != "0"
os.environ.setdefault(_FLAG, "0")


def hip_graphs_safe() -> bool:
    """True when HIP-graph replay of large graphs can be trusted in this process: the packet-capture switch is off and
    was in the environment before the HIP runtime could have been loaded."""
    return os.environ.get(_FLAG) == "0" and not _torch_loaded_first
