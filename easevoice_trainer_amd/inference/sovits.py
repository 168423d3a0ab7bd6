"""Loading a trained s2 export for the inference-side entry points.

Mirrors TTS.init_vits_weights (src/easevoice/inference/tts.py:265-299): the export written by SovitsTrain._save_epoch is
`{"weight": fp16 state_dict without enc_q.*, "config": hps, "info": ...}`; the model is rebuilt from `config`, enc_q is
dropped, the weights are loaded non-strictly and the module is put in eval mode.  On top of that the conv weight bank of
the HIP kernels (runtime.ModelRuntime) is built and folded once, since inference never updates the parameters."""
import torch

from ..hip import lib as L
from ..module import models
from ..runtime import ModelRuntime


class SoVITSVoice:
    """`decode(codes, text, refer, noise_scale, speed)` and `extract_latent(ssl)` of SynthesizerTrn (models.py:974-1018)
    on a loaded export."""

    def __init__(self, weights, device="cuda:0", dtype=torch.bfloat16):
        ck = torch.load(weights, map_location="cpu", weights_only=False) if isinstance(weights, str) else weights
        hps = ck["config"]
        if hasattr(hps, "model_dump"):
            hps = hps.model_dump()
        if ck["weight"]["enc_p.text_embedding.weight"].shape[0] == 322:
            raise ValueError("The model is version v1, please use the latest version model.")
        self.hps = hps
        d = hps["data"]
        self.sampling_rate, self.hop_length = d["sampling_rate"], d["hop_length"]
        self.filter_length, self.win_length = d["filter_length"], d["win_length"]
        net = models.SynthesizerTrn(d["filter_length"] // 2 + 1, hps["train"]["segment_size"] // d["hop_length"],
                                    n_speakers=d["n_speakers"], **hps["model"])
        del net.enc_q                      # not part of the export, not used by decode
        net.load_state_dict({k: v.float() for k, v in ck["weight"].items()}, strict=False)
        net.eval()
        self.rt = ModelRuntime(net, dtype=dtype, device=device)
        self.rt.prepare(force=True)
        self.model, self.device, self.dtype = net, torch.device(device), dtype

    @torch.no_grad()
    def decode(self, codes, text, refer, noise_scale=0.5, speed=1, noise=None):
        L.set_half(self.dtype)
        mv = lambda t: t.to(self.device)
        refer = [mv(r) for r in refer] if isinstance(refer, (list, tuple)) else mv(refer)
        return self.model.decode(mv(codes), mv(text), refer, noise_scale=noise_scale, speed=speed,
                                 noise=None if noise is None else mv(noise))

    @torch.no_grad()
    def extract_latent(self, ssl):
        L.set_half(self.dtype)
        return self.model.extract_latent(ssl.to(self.device))
