"""Upper bound of what hiding the parameter bookkeeping could buy: the s2 bench step with the three bookkeeping launches
(weight-norm gradient, AdamW, weight-norm fold) left out after warm-up.  The numbers of such a run are NOT a training
result; only ms_per_step against the normal run on the same box means something.
usage: python tools/exp/nobook.py [bench.py args]"""
import os
import runpy
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import easevoice_trainer_amd.runtime as R          # noqa: E402
from easevoice_trainer_amd.hip import conv as HC  # noqa: E402

which = os.environ.get("NOBOOK", "grad,adamw,fold").split(",")
if "adamw" in which:
    def _step(self, grad_scale=1.0):
        self.step_count += 1
        self.arena.updates += 1
    R.FlatAdamW.step = _step
if "grad" in which:
    def _grads(self, lo=None, hi=None):
        self.join_side()
    HC.WeightBank.grads = _grads
if "fold" in which:
    _fold = HC.WeightBank.fold

    def _fold_few(self):
        n = getattr(self, "_nfold", 0)
        if n < 2:
            _fold(self)
        self._nfold = n + 1
    HC.WeightBank.fold = _fold_few
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "bench.py"),
               run_name="__main__")
