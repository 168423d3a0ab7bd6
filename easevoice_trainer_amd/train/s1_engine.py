"""The s1 training micro-step (Text2SemanticLightningModule.training_step,
src/easevoice/soundstorm/auto_reg/models/t2s_lightning_module.py:41-89) without Lightning: manual optimisation,
gradients accumulated in the flat arena, ScaledAdam + the pinned-LR schedule, optimiser step on
`batch_idx > 0 and batch_idx % 4 == 0` exactly like the reference.  Data-parallel: ONE all-reduce of the flat gradient
arena per optimiser step (the reference all-reduces on each of the accumulation micro-batches)."""
import torch

from ..auto_reg.optim import ScaledAdam, WarmupCosineLRSchedule
from ..auto_reg.t2s_model import Text2SemanticDecoder
from ..hip import lib as L
from ..runtime import ParamArena


class S1Engine:
    def __init__(self, config: dict, device="cuda:0", dtype=torch.bfloat16, reducer=None):
        L.lib()
        self.config, self.device, self.dtype, self.reducer = config, torch.device(device), dtype, reducer
        self.model = Text2SemanticDecoder(config=config, top_k=3).to(self.device)
        self.model.cd = dtype
        # the vocabulary projection (1025 x 512) is run as a 1152-row GEMM: its zero padding rows live behind the parameter
        from ..hip.linear import N_ALIGN
        w = self.model.ar_predict_layer.weight
        pad_rows = (w.size(0) + N_ALIGN - 1) // N_ALIGN * N_ALIGN
        self.arena = ParamArena(self.model, self.device, reserve={"ar_predict_layer.weight": pad_rows * w.size(1)})
        self.bank = self.model.attach_bank(dtype, self.device)
        o = config["optimizer"]
        self.optimizer = ScaledAdam(self.arena, lr=0.01, betas=(0.9, 0.95), clipping_scale=2.0,
                                    clipping_update_period=1000)
        self.scheduler = WarmupCosineLRSchedule(self.optimizer, init_lr=o["lr_init"], peak_lr=o["lr"],
                                                end_lr=o["lr_end"], warmup_steps=o["warmup_steps"],
                                                total_steps=o["decay_steps"])
        self.arena.zero_grad()
        self._views = [(p, p.grad) for p in self.model.parameters()]
        for p, v in self._views:       # the GEMM weight-gradient launches accumulate straight into these (hip/linear.py)
            p._evt_grad_view = v

    def micro_step(self, batch: dict, batch_idx: int):
        """one micro-batch: forward_old (or the DPO `forward` when config train.if_dpo, t2s_lightning_module.py:44) +
        backward (+ optimiser step on the reference's schedule)"""
        fwd = self.model.forward if self.config.get("train", {}).get("if_dpo", False) is True else self.model.forward_old
        loss, acc = fwd(batch["phoneme_ids"], batch["phoneme_ids_len"], batch["semantic_ids"],
                        batch["semantic_ids_len"], batch["bert_feature"])
        # autograd keeps the produced gradient tensors (no per-parameter `grad += new` launch); they are accumulated
        # into the flat arena with one multi-tensor add
        for p, _v in self._views:
            p.grad = None
        loss.backward()
        dst, src = [], []
        for p, v in self._views:
            if p.grad is not None:
                dst.append(v)
                src.append(p.grad if p.grad.dtype == v.dtype else p.grad.to(v.dtype))
            p.grad = v
        if dst:
            torch._foreach_add_(dst, src)
        stepped = False
        if batch_idx > 0 and batch_idx % 4 == 0:
            if self.reducer is not None:
                self.reducer.all_reduce(self.arena.grad)
                self.arena.grad.mul_(1.0 / self.reducer.world)
            self.optimizer.step()
            self.bank.mark_dirty()        # weights written through raw pointers: refold the GEMM images on next use
            self.optimizer.zero_grad()
            self.scheduler.step()
            stepped = True
        return loss.detach(), acc.detach(), stepped
