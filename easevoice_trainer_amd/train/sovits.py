"""SovitsTrain — the reference's s2 trainer surface (src/train/sovits.py:37-211,213-624) on the MI355X engine.

Same params dataclass, configs/s2.json, output directory / checkpoint / export layout and stdout loss lines; the step
itself is train/s2_engine.py.  One process per GPU: when launched under torch.distributed.run (WORLD_SIZE > 1) the
gradient arenas are all-reduced over RCCL; the reference hard-codes a single process (sovits.py:199-210)."""
import json
import logging
import os
from collections import OrderedDict
from dataclasses import dataclass

import torch

from ..dist import GradReducer, init_process_group_from_env
from ..utils import ckpt, tb
from ..utils.connector import MultiProcessOutputConnector
from .data import SyntheticS2Batches, open_source
from .helper import TrainOutput, default_pretrained, get_sovits_train_dir, repo_root, train_logs_path
from .s2_engine import S2Engine

logger = logging.getLogger("easevoice")


@dataclass
class SovitsTrainParams:
    batch_size: int = 12
    total_epochs: int = 8
    text_low_lr_rate: float = 0.4
    pretrained_s2G: str = ""
    pretrained_s2D: str = ""
    if_save_latest: bool = True
    if_save_every_weights: bool = True
    save_every_epoch: int = 5
    gpu_ids: str = "0"
    train_input_dir: str = ""
    output_model_name: str = ""
    project_dir: str = ""


def compute_dtype(hps):
    """The step's compute type from the config, as the reference reads it (src/train/sovits.py:378,459: `fp16_run` switches the
    float16 autocast and the GradScaler on): fp16_run true -> torch.float16 on the IEEE-half build of the library with the
    device-side GradScaler (train/s2_engine.py), false -> float32.  EVT_HALF=bf16 runs a half-precision config in bfloat16
    instead -- same kernels and speed, fp32's exponent range, so no loss scaling and no skipped steps; a deliberate departure
    from the reference that has to be asked for (until round 5 it was the silent default)."""
    if not hps["train"].get("fp16_run", False):
        return torch.float32
    half = os.environ.get("EVT_HALF", "f16").lower()
    if half in ("f16", "fp16", "float16", "half"):
        return torch.float16
    if half in ("bf16", "bfloat16"):
        return torch.bfloat16
    raise ValueError(f"EVT_HALF={half!r}: expected f16 or bf16")


class SovitsTrain:
    def __init__(self, params: SovitsTrainParams, dtype=None, config_path=None):
        hps = json.load(open(config_path or os.path.join(repo_root(), "configs", "s2.json")))
        t = hps["train"]
        t["batch_size"], t["epochs"], t["text_low_lr_rate"] = params.batch_size, params.total_epochs, params.text_low_lr_rate
        t["if_save_latest"], t["if_save_every_weights"] = params.if_save_latest, params.if_save_every_weights
        t["save_every_epoch"], t["gpu_numbers"] = params.save_every_epoch, params.gpu_ids
        t.setdefault("log_interval", 10)
        hps["name"] = params.output_model_name
        hps["data"]["exp_dir"] = params.train_input_dir
        t["output_dir"] = get_sovits_train_dir(params.project_dir, params.output_model_name)
        t["train_logs_dir"] = os.path.join(t["output_dir"], train_logs_path)
        t["save_weight_dir"] = t["output_dir"]
        # an empty path (or the relative default the reference's UI sends) means "the stock pretrained pair", sovits.py:149-157
        stock = ("", "pretrained/gsv-v2final-pretrained/s2G2333k.pth", "pretrained/gsv-v2final-pretrained/s2D2333k.pth")
        t["pretrained_s2G"] = default_pretrained("s2G") if params.pretrained_s2G in stock[:2] else params.pretrained_s2G
        t["pretrained_s2D"] = default_pretrained("s2D") if params.pretrained_s2D in (stock[0], stock[2]) else params.pretrained_s2D
        os.makedirs(t["output_dir"], exist_ok=True)
        os.makedirs(t["train_logs_dir"], exist_ok=True)
        self.hps, self.params, self.dtype = hps, params, (dtype if dtype is not None else compute_dtype(hps))
        self.global_step = 0

    # ---- export for inference (sovits.py:179-196): fp16 weights without enc_q, config, info ----
    def _save_epoch(self, state_dict, name, epoch, steps):
        opt = OrderedDict()
        opt["weight"] = {k: v.detach().cpu().half() for k, v in state_dict.items() if "enc_q" not in k}
        opt["config"] = self.hps
        opt["info"] = "%sepoch_%siteration" % (epoch, steps)
        path = os.path.join(self.hps["train"]["save_weight_dir"], f"{name}.pth")
        ckpt.save_with_torch(opt, path)
        return path

    def train(self):
        world, rank, local = init_process_group_from_env(gpu_ids=self.params.gpu_ids)
        self._run(rank, world, local)
        return TrainOutput(model_path=self.hps["train"]["output_dir"])

    @staticmethod
    def _device(local):
        """the rank's GPU (the only place the trainer names a device)"""
        device = torch.device("cuda", local)
        torch.cuda.set_device(device)
        return device

    def _run(self, rank, world, local):
        hps, t = self.hps, self.hps["train"]
        torch.manual_seed(t["seed"])
        device = self._device(local)
        reducer = GradReducer(world) if world > 1 else None
        eng = S2Engine(hps, device, self.dtype, reducer=reducer)
        optim_g, optim_d = eng.build_optimizers()
        if os.environ.get("EVT_GRAPHS", "1") != "0":
            # batches whose shapes repeat (bucketed / fixed-length sources) are replayed as HIP graphs after two eager
            # steps of that shape; other shapes keep running eagerly
            # (up to EVT_GRAPH_SHAPES of them, default 64; the engine also stops capturing when device memory runs short)
            eng.enable_graphs(warmup_steps=2, max_shapes=int(os.environ.get("EVT_GRAPH_SHAPES", "64")))
        source = open_source("s2", hps["data"]["exp_dir"], device,
                             lambda n: SyntheticS2Batches(t["batch_size"], 4, n, device, seed=t["seed"], rank=rank, world=world),
                             batch_size=t["batch_size"], cfg=hps["data"], rank=rank, world=world)
        if int(getattr(source, "pad_frames", 0) or 0) > 0:
            # the reader pads time axes beyond the reference's collate length: the one consumer that would notice is told
            eng.net_g.ref_enc.mask_beyond_collate = True
        # resume, else pretrained (sovits.py:327-366)
        try:
            _, _, _, epoch_str = ckpt.load_checkpoint(ckpt.latest_checkpoint_path(t["train_logs_dir"], "D_*.pth"),
                                                      eng.net_d, optim_d)
            _, _, _, epoch_str = ckpt.load_checkpoint(ckpt.latest_checkpoint_path(t["train_logs_dir"], "G_*.pth"),
                                                      eng.net_g, optim_g)
            self.global_step = (epoch_str - 1) * len(source)
        except Exception as e:
            logger.warning(f"load failed, exception: {e}, use pretrained instead")
            epoch_str, self.global_step = 1, 0
            if t["pretrained_s2G"] and os.path.exists(t["pretrained_s2G"]):
                eng.net_g.load_state_dict(torch.load(t["pretrained_s2G"], map_location="cpu", weights_only=False)["weight"],
                                          strict=False)
            if t["pretrained_s2D"] and os.path.exists(t["pretrained_s2D"]):
                eng.net_d.load_state_dict(torch.load(t["pretrained_s2D"], map_location="cpu", weights_only=False)["weight"])
        cb = eng.net_g.quantizer.vq.layers[0]._codebook
        if float(cb.inited.float().cpu()) == 0.0 and os.environ.get("EVT_SYNTHETIC_STEPS"):
            cb.embed.normal_()          # synthetic runs: a random codebook instead of k-means over noise features
            cb.inited.fill_(1.0)
        # (no pretrained generator and real data: the codebook is k-means-initialised from the first batch, as in the
        # reference, core_vq.py:140-149 -- module/models.py:_Codebook.init_embed_)
        if reducer is not None:
            reducer.broadcast_params(eng.rt_g.arena.param)
            reducer.broadcast_params(eng.rt_d.arena.param)
        # ExponentialLR(gamma=lr_decay), fast-forwarded epoch_str times, stepped once per epoch (sovits.py:368-376,409)
        def set_lr(epoch_index):
            f = t["lr_decay"] ** epoch_index
            for opt in (optim_g, optim_d):
                for g in opt.param_groups:
                    g["lr"] = g["initial_lr"] * f
        sched_epoch = epoch_str
        set_lr(sched_epoch)
        connector = MultiProcessOutputConnector()
        writer = tb.open_writer(tb.tensorboard_log_dir(hps["name"])) if rank == 0 else None   # sovits.py:220
        for epoch in range(epoch_str, t["epochs"] + 1):
            source.set_epoch(epoch)
            for batch_idx, (ssl, _ssl_len, spec, spec_len, y, _y_len, text, text_len) in enumerate(source):
                out = eng.step(ssl, spec, spec_len, y, text, text_len)
                if self.global_step % 10 == 0 and rank == 0:       # one host sync every 10 steps, sovits.py:527-536
                    loss = float(out.gen_all)
                    connector.write_loss(self.global_step, loss=loss, other={
                        "loss/g/total": loss, "loss/d/total": float(out.disc),
                        "learning_rate": optim_g.param_groups[0]["lr"]})
                if self.global_step % 5 == 0 and writer is not None:     # the scalars of sovits.py:539-565
                    try:
                        tb.log_scalars(writer, self.global_step, {
                            "loss/g/total": float(out.gen_all), "loss/d/total": float(out.disc),
                            "learning_rate": optim_g.param_groups[0]["lr"],
                            "grad_norm_d": float(out.grad_sumsq_d) ** 0.5, "grad_norm_g": float(out.grad_sumsq_g) ** 0.5,
                            "loss/g/fm": float(out.fm), "loss/g/mel": float(out.mel),
                            "loss/g/kl_ssl": float(out.kl_ssl), "loss/g/kl": float(out.kl)})
                    except Exception as e:      # summaries never take a training run down
                        logger.warning(f"tensorboard summary skipped: {e}")
                        writer = None
                self.global_step += 1
            if epoch % t["save_every_epoch"] == 0 and rank == 0:
                tag = "latest" if t["if_save_latest"] else str(self.global_step)
                ckpt.save_checkpoint(eng.net_g, optim_g, t["learning_rate"], epoch,
                                     os.path.join(t["train_logs_dir"], f"G_{tag}.pth"))
                ckpt.save_checkpoint(eng.net_d, optim_d, t["learning_rate"], epoch,
                                     os.path.join(t["train_logs_dir"], f"D_{tag}.pth"))
                if t["if_save_every_weights"]:
                    self._save_epoch(eng.net_g.state_dict(), hps["name"] + f"_e{epoch}_s{self.global_step}", epoch,
                                     self.global_step)
            sched_epoch += 1
            set_lr(sched_epoch)
        self.engine = eng
