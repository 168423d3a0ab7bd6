"""TEST INFRASTRUCTURE ONLY — never imported by the product path.

Makes the reference's own Python modules importable from /root/reference (read-only)
inside the build container so golden vectors can be generated from the reference's
*code*, not from a restatement.  SURVEY.md §8(c): three third-party packages are absent
here and get minimal stand-ins:

  * librosa.filters.mel   -> oracle.melbank.slaney_mel (librosa==0.9.2 formula; call
                             sites mel_processing.py:82-84,106-108)
  * torchmetrics.classification.MulticlassAccuracy -> metric only, not in the loss
  * ffmpeg                -> empty (data_utils import only)

/root/reference does not exist on the GPU box; nothing under tests -m gpu, smoke() or
bench.py may call this module.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("EVT_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "src", "easevoice"))


def install() -> None:
    """Inject shims and put the reference on sys.path (idempotent)."""
    if not reference_available():
        raise RuntimeError(f"reference not present at {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True
    os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    if "librosa" not in sys.modules:
        from oracle.melbank import slaney_mel

        librosa = types.ModuleType("librosa")
        filters = types.ModuleType("librosa.filters")

        def mel(sr, n_fft, n_mels=128, fmin=0.0, fmax=None, **kw):
            return slaney_mel(sr, n_fft, n_mels, fmin, fmax)

        filters.mel = mel
        librosa.filters = filters
        sys.modules["librosa"] = librosa
        sys.modules["librosa.filters"] = filters

    if "torchmetrics" not in sys.modules:
        import torch

        tm = types.ModuleType("torchmetrics")
        cls = types.ModuleType("torchmetrics.classification")

        class MulticlassAccuracy(torch.nn.Module):
            def __init__(self, num_classes, top_k=1, average="micro",
                         multidim_average="global", ignore_index=None):
                super().__init__()
                self.top_k, self.ignore_index = top_k, ignore_index

            def forward(self, logits, target):  # logits [B,C,T], target [B,T]
                topk = logits.topk(self.top_k, dim=1).indices
                hit = (topk == target.unsqueeze(1)).any(dim=1)
                keep = torch.ones_like(hit) if self.ignore_index is None else target != self.ignore_index
                return (hit & keep).sum().float() / keep.sum().clamp(min=1).float()

        cls.MulticlassAccuracy = MulticlassAccuracy
        tm.classification = cls
        sys.modules["torchmetrics"] = tm
        sys.modules["torchmetrics.classification"] = cls

    if "ffmpeg" not in sys.modules:
        sys.modules["ffmpeg"] = types.ModuleType("ffmpeg")
