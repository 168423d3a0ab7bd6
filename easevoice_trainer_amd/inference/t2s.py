"""Loading a trained s1 export for decoding, the way TTS.init_t2s_weights does (src/easevoice/inference/tts.py:301-314):
`{"weight": {"model.<key>": half tensor}, "config": yaml dict, "info": ...}` as written by GPTTrain / the reference's
GPTCheckpoint; `max_sec` comes from the export's config, the semantic rate is 50 tokens per second there (`hz = 50`)."""
import torch

from ..auto_reg.t2s_model import Text2SemanticDecoder


class T2SVoice:
    def __init__(self, weights, device="cuda:0", dtype=torch.bfloat16):
        ck = torch.load(weights, map_location="cpu", weights_only=False) if isinstance(weights, str) else weights
        self.config = ck["config"]
        self.hz, self.max_sec = 50, self.config["data"]["max_sec"]
        model = Text2SemanticDecoder(config=self.config, top_k=3)
        model.load_state_dict({k[len("model."):]: v.float() for k, v in ck["weight"].items() if k.startswith("model.")})
        self.model = model.to(device).eval()
        self.model.cd = dtype
        self.device = torch.device(device)

    @property
    def early_stop_num(self):
        return self.hz * self.max_sec
