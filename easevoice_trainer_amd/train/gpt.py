"""GPTTrain — the reference's s1 trainer surface (src/train/gpt.py:27-195 + t2s_lightning_module.py:41-122) without
pytorch_lightning: same params, configs/gpt.yaml, `<out>/logs/ckpt/epoch=E-step=S.ckpt` resume files (state_dict keys
carry Lightning's `model.` prefix), `<out>/<name>-e{E}.ckpt` half-precision exports {"weight", "config", "info"} that
src/easevoice/inference/tts.py:301-315 loads, and one `loss-of-easevoice` line per micro-step on rank 0."""
import logging
import os
import re
from collections import OrderedDict
from dataclasses import dataclass

import torch
import yaml

from ..dist import GradReducer, init_process_group_from_env
from ..utils import ckpt, tb
from ..utils.connector import MultiProcessOutputConnector
from .data import SyntheticS1Batches, open_source
from .helper import TrainOutput, default_pretrained, get_gpt_train_dir, repo_root, train_logs_path
from .s1_engine import S1Engine

logger = logging.getLogger("easevoice")


@dataclass
class GPTTrainParams:
    batch_size: int = 12
    total_epochs: int = 15
    save_every_epoch: int = 5
    if_dpo: bool = False
    if_save_latest: bool = True
    if_save_every_weights: bool = True
    gpu_ids: str = "0"
    model_path: str = ""
    train_input_dir: str = ""
    output_model_name: str = ""
    project_dir: str = ""


def compute_dtype(config):
    """train.precision as Lightning reads it (src/train/gpt.py:112-113,157; configs/gpt.yaml: "16-mixed"): "16-mixed" ->
    torch.float16 with loss scaling (train/s1_engine.py), "bf16-mixed" -> bfloat16, "32" -> float32.  EVT_HALF=bf16 runs a
    16-mixed config in bfloat16 instead (no loss scaling needed; the default before round 5)."""
    prec = str(config["train"].get("precision", "32"))
    if prec.startswith("bf16"):
        return torch.bfloat16
    if prec.startswith("16"):
        half = os.environ.get("EVT_HALF", "f16").lower()
        if half in ("bf16", "bfloat16"):
            return torch.bfloat16
        if half in ("f16", "fp16", "float16", "half"):
            return torch.float16
        raise ValueError(f"EVT_HALF={half!r}: expected f16 or bf16")
    return torch.float32


class GPTTrain:
    def __init__(self, params: GPTTrainParams, dtype=None, config_path=None):
        self.config = yaml.safe_load(open(config_path or os.path.join(repo_root(), "configs", "gpt.yaml")))
        self.params, self.dtype = params, (dtype if dtype is not None else compute_dtype(self.config))
        self.train_output = get_gpt_train_dir(params.project_dir, params.output_model_name)
        self.train_logs_output = os.path.join(self.train_output, train_logs_path)
        self.train_ckpts_output = os.path.join(self.train_logs_output, "ckpt")
        for d in (self.train_output, self.train_logs_output, self.train_ckpts_output):
            os.makedirs(d, exist_ok=True)
        c = self.config["train"]
        c["batch_size"], c["epochs"], c["save_every_n_epoch"] = params.batch_size, params.total_epochs, params.save_every_epoch
        c["if_dpo"], c["if_save_latest"], c["if_save_every_weights"] = params.if_dpo, params.if_save_latest, params.if_save_every_weights
        c["half_weights_save_dir"], c["output_name"] = self.train_output, params.output_model_name
        # the reference's dataclass default is its stock pretrained checkpoint (src/train/gpt.py:35)
        self.config["pretrained_s1"] = params.model_path or default_pretrained("s1")
        self.config["logs_output_dir"] = self.train_logs_output
        self.global_step = 0

    @staticmethod
    def _get_newest_ckpt(file_list):
        info = []
        for s in file_list or []:
            m = re.match(r"epoch=(\d+)-step=(\d+)\.ckpt", s)
            if m:
                info.append((int(m.group(1)), int(m.group(2)), s))
        return sorted(info, reverse=True)[0][2] if info else None

    def _export(self, model, epoch):
        od = OrderedDict()
        od["weight"] = OrderedDict(("model." + k, v.detach().cpu().half()) for k, v in model.state_dict().items())
        od["config"] = self.config
        od["info"] = "GPT-e%s" % epoch
        path = os.path.join(self.train_output, "%s-e%s.ckpt" % (self.params.output_model_name, epoch))
        torch.save(od, path)
        return path

    @staticmethod
    def _device(local):
        """the rank's GPU (the only place the trainer names a device)"""
        device = torch.device("cuda", local)
        torch.cuda.set_device(device)
        return device

    def train(self):
        world, rank, local = init_process_group_from_env(gpu_ids=self.params.gpu_ids)
        cfg, c = self.config, self.config["train"]
        torch.manual_seed(c["seed"])
        device = self._device(local)
        reducer = GradReducer(world) if world > 1 else None
        eng = S1Engine(cfg, device, self.dtype, reducer=reducer)
        if cfg.get("pretrained_s1") and os.path.exists(cfg["pretrained_s1"]):
            w = torch.load(cfg["pretrained_s1"], map_location="cpu", weights_only=False)["weight"]
            eng.model.load_state_dict({k[len("model."):]: v.float() for k, v in w.items() if k.startswith("model.")})
        start_epoch = 0
        newest = self._get_newest_ckpt(os.listdir(self.train_ckpts_output))
        if newest:
            ck = torch.load(os.path.join(self.train_ckpts_output, newest), map_location="cpu", weights_only=False)
            eng.model.load_state_dict({k[len("model."):]: v for k, v in ck["state_dict"].items()})
            opt_sd = ck["optimizer_states"][0]
            if "state" in opt_sd and "param_groups" in opt_sd:
                # a checkpoint the reference's Lightning trainer wrote (src/train/gpt.py:172-177): its optimiser's own
                # state_dict, parameters in the order of the module's state_dict
                ours = set(eng.optimizer.names)
                order = [k[len("model."):] for k in ck["state_dict"] if k.startswith("model.") and k[len("model."):] in ours]
                eng.optimizer.load_reference_state(opt_sd, order)
            else:
                eng.optimizer.load_state_dict({k: (v.to(device) if torch.is_tensor(v) else v) for k, v in opt_sd.items()})
            if ck.get("lr_schedulers"):
                eng.scheduler.load_state_dict(ck["lr_schedulers"][0])
            amp = ck.get("MixedPrecisionPlugin") or ck.get("MixedPrecision")       # Lightning 2.0 / later 2.x key
            if amp and eng.scaler.enabled:
                eng.scaler.load_state_dict(amp)
            start_epoch, self.global_step = ck["epoch"] + 1, ck["global_step"]
        if reducer is not None:
            reducer.broadcast_params(eng.arena.param)
        source = open_source("s1", self.params.train_input_dir, device,
                             lambda n: SyntheticS1Batches(c["batch_size"], 256, 768, n, device, seed=c["seed"], rank=rank),
                             batch_size=c["batch_size"], cfg=dict(cfg["data"], if_dpo=c.get("if_dpo", False)),
                             rank=rank, world=world)
        connector = MultiProcessOutputConnector()
        step_no = 0
        writer = None
        if rank == 0:       # TensorBoardLogger(name=<model name>, save_dir=<tb_logs>) of src/train/gpt.py:143: version_<n> below it
            base = tb.tensorboard_log_dir(self.params.output_model_name or "gpt")
            n = 0
            while os.path.isdir(os.path.join(base, f"version_{n}")):
                n += 1
            writer = tb.open_writer(os.path.join(base, f"version_{n}"))
        for epoch in range(start_epoch, c["epochs"]):
            source.set_epoch(epoch)
            for batch_idx, batch in enumerate(source):
                loss, acc, stepped = eng.micro_step(batch, batch_idx)
                self.global_step += int(stepped)
                if rank == 0:
                    lv, av, lr = float(loss), float(acc), eng.scheduler.get_last_lr()[0]
                    connector.write_loss(step_no, loss=lv, other={"acc": av, "lr": lr, "epoch": epoch})
                    # the three self.log() scalars of t2s_lightning_module.py:58-80 (Lightning's per-step tag names)
                    tb.log_scalars(writer, step_no, {"total_loss_step": lv, "lr": lr, "top_3_acc_step": av})
                step_no += 1
            if (epoch + 1) % c["save_every_n_epoch"] == 0 and rank == 0:
                # save first, then drop the files that were there before (src/train/gpt.py:66-78): a failed write leaves
                # the previous resume point in place
                before = os.listdir(self.train_ckpts_output) if c["if_save_latest"] else []
                sd = OrderedDict(("model." + k, v.detach().cpu().clone()) for k, v in eng.model.state_dict().items())
                # optimiser and scheduler state in the layout the reference's Lightning checkpoint holds them in
                # (torch.optim.Optimizer.state_dict() of its batched ScaledAdam, parameters in named_parameters() order;
                # the scheduler's attribute dict), so that either trainer can pick the other's file up.  Lightning's own
                # "loops" / "callbacks" entries are not written (its legacy path restores epoch / global_step without them).
                order = [n for n, _ in eng.model.named_parameters() if n in set(eng.optimizer.names)]
                opt_sd = eng.optimizer.reference_state_dict(order)
                sch_sd = eng.scheduler.state_dict()
                new_name = f"epoch={epoch}-step={self.global_step}.ckpt"
                blob = {"epoch": epoch, "global_step": self.global_step, "state_dict": sd,
                        "optimizer_states": [opt_sd], "lr_schedulers": [sch_sd], "hyper_parameters": {"config": cfg}}
                if eng.scaler.enabled:
                    # precision 16-mixed: Lightning 2.0 stores its AMP plugin's state -- the GradScaler's state_dict --
                    # under the plugin's class name (trainer/connectors/checkpoint_connector.py: dump_checkpoint)
                    blob["MixedPrecisionPlugin"] = eng.scaler.state_dict()
                ckpt.save_with_torch(blob, os.path.join(self.train_ckpts_output, new_name))
                for name in before:
                    if name != new_name:
                        try:
                            os.remove(os.path.join(self.train_ckpts_output, name))
                        except OSError:
                            pass
                if c["if_save_every_weights"]:
                    self._export(eng.model, epoch + 1)
        self.engine = eng
        return TrainOutput(model_path=self.train_output)
