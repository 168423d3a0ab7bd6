"""The s1 training micro-step (Text2SemanticLightningModule.training_step,
src/easevoice/soundstorm/auto_reg/models/t2s_lightning_module.py:41-89) without Lightning: manual optimisation,
gradients accumulated in the flat arena, ScaledAdam + the pinned-LR schedule, optimiser step on
`batch_idx > 0 and batch_idx % 4 == 0` exactly like the reference.  Data-parallel: the flat gradient arena is reduced
once per optimiser step (the reference all-reduces on each of the accumulation micro-batches), in pieces that start
while the last micro-batch's backward is still running: the arena is in parameter order (embeddings, blocks 0..23,
predict layer), the backward walks the blocks from 23 down, so when the gradient reaches the input of block k the
ranges of blocks k+1.. and of the predict layer are final -- they go to the side stream (sum + the 1/world scale) and
the rest follows after the backward (EVT_DP_S1_CUTS: comma-separated block indices, default "16,8"; "" = one piece)."""
import os

import torch

from ..auto_reg.optim import ScaledAdam, WarmupCosineLRSchedule
from ..auto_reg.t2s_model import Text2SemanticDecoder
from ..hip import lib as L
from ..runtime import DeviceGradScaler, ParamArena


class S1Engine:
    def __init__(self, config: dict, device="cuda:0", dtype=torch.bfloat16, reducer=None, scaler_args=None):
        """dtype torch.float16: the reference's `precision: 16-mixed` (configs/gpt.yaml:6; Lightning's AMP plugin around the
        manual optimisation of t2s_lightning_module.py:41-89: manual_backward scales the loss, the optimiser step every fourth
        micro-batch unscales, skips on an overflow and updates the scale).  The scaled gradients of the four micro-batches
        accumulate in the arena under ONE scale (update() only runs at the step), the arena is unscaled and checked once, after
        the data-parallel sum; the skip decision is read on the host there -- one read per four micro-batches, exactly where
        torch's GradScaler.step has its own `.item()`."""
        L.set_half(dtype)
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
        if os.environ.get("EVT_S1_WGRAD_SIDE", "1") != "0":       # the dense layers' dW launches next to the backward chain
            self.bank.enable_side_stream()
        o = config["optimizer"]
        self.optimizer = ScaledAdam(self.arena, lr=0.01, betas=(0.9, 0.95), clipping_scale=2.0,
                                    clipping_update_period=1000)
        self.scheduler = WarmupCosineLRSchedule(self.optimizer, init_lr=o["lr_init"], peak_lr=o["lr"],
                                                end_lr=o["lr_end"], warmup_steps=o["warmup_steps"],
                                                total_steps=o["decay_steps"])
        self.scaler = DeviceGradScaler(self.device, enabled=dtype == torch.float16, **(scaler_args or {}))
        self.skipped_steps = 0
        self.arena.zero_grad()
        self._views = [(p, p.grad) for p in self.model.parameters()]
        for p, v in self._views:       # the GEMM weight-gradient launches accumulate straight into these (hip/linear.py)
            p._evt_grad_view = v
        # overlapped reduction: cut points (block indices, descending) and where each block's parameters start in the arena
        cuts = os.environ.get("EVT_DP_S1_CUTS", "16,8")
        nl = len(self.model.h.layers)
        self._cuts = sorted({int(c) for c in cuts.split(",") if c.strip() != "" and 0 <= int(c) < nl - 1}, reverse=True)
        names = [n for n, _ in self.model.named_parameters()]
        self._block_start = {}
        for i in range(nl):
            first = next(n for n in names if n.startswith(f"h.layers.{i}."))
            self._block_start[i] = self.arena.offsets[first]
        self._param_off = [self.arena.offsets[n] for n in names]
        self._reduced_from = None      # arena offset from which the gradients are already with the side stream

    def micro_step(self, batch: dict, batch_idx: int):
        """one micro-batch: forward_old (or the DPO `forward` when config train.if_dpo, t2s_lightning_module.py:44) +
        backward (+ optimiser step on the reference's schedule)"""
        L.set_half(self.dtype)
        dpo = self.config.get("train", {}).get("if_dpo", False) is True
        fwd = self.model.forward if dpo else self.model.forward_old
        stepping = batch_idx > 0 and batch_idx % 4 == 0
        # the DPO forward walks the block stack twice (chosen and rejected targets, t2s_model.py:393-429): a hook at a cut
        # block's input would fire in the rejected pass's backward while the chosen pass still has gradients to add to the
        # same arena range -> a range is only final after the whole backward, so DPO reduces the arena in one piece
        overlap = stepping and not dpo and self.reducer is not None and self.reducer.active and bool(self._cuts)
        # the forward plants the tensor hooks at the cut blocks' inputs
        self.model.h.grad_hook_blocks = tuple(self._cuts) if overlap else ()
        try:
            loss, acc = fwd(batch["phoneme_ids"], batch["phoneme_ids_len"], batch["semantic_ids"],
                            batch["semantic_ids_len"], batch["bert_feature"])
        finally:
            self.model.h.grad_hook_blocks = ()
        # autograd keeps the produced gradient tensors (no per-parameter `grad += new` launch); they are accumulated
        # into the flat arena with one multi-tensor add (per reduced piece when the reduction overlaps the backward)
        for p, _v in self._views:
            p.grad = None
        self._reduced_from = None
        self.model.h.grad_hook = self._piece_done if overlap else None
        self.bank.side_begin()         # this backward's dense-layer dW launches run next to its chain (hip/linear.py)
        try:
            self.scaler.scale(loss).backward()
        finally:
            self.model.h.grad_hook = None
            self.bank.side_end()       # joined: the weight gradients of the dense layers are in the arena from here on
        hi = self._reduced_from if self._reduced_from is not None else self.arena.grad.numel()
        self._gather(0, hi)
        for p, v in self._views:
            p.grad = v
        stepped = False
        if stepping:
            self._finish_window(hi)
            stepped = True
        return loss.detach(), acc.detach(), stepped

    def _finish_window(self, hi=None):
        """the optimiser step of an accumulation window on the gradients in the arena (t2s_lightning_module.py:53-56 under
        the AMP plugin: scaler.step = unscale, skip on inf / nan, step; scaler.update; zero_grad; scheduler.step)"""
        hi = self.arena.grad.numel() if hi is None else hi
        if self.reducer is not None and self.reducer.active:
            self.reducer.all_reduce(self.arena.grad[:hi], async_op=self.arena.grad.is_cuda, average=True)
            self.reducer.wait()
        overflow = False
        if self.scaler.enabled:
            self.scaler.unscale_(self)                                    # self.arena: every rank holds the same sums
            overflow = bool(self.scaler.found_inf(self).item() != 0.0)    # GradScaler.step's host read
        if not overflow:
            self.optimizer.step()
            self.bank.mark_dirty()    # weights written through raw pointers: refold the GEMM images on next use
        else:
            self.skipped_steps += 1
        self.scaler.update()
        self.optimizer.zero_grad()
        self.scheduler.step()         # Lightning steps the scheduler whether or not the scaler skipped the optimiser

    def _gather(self, lo, hi):
        """add the gradient tensors autograd produced for the parameters in arena range [lo, hi) into their arena views"""
        dst, src = [], []
        for (p, v), o in zip(self._views, self._param_off):
            if lo <= o < hi and p.grad is not None and p.grad is not v:
                dst.append(v)
                src.append(p.grad if p.grad.dtype == v.dtype else p.grad.to(v.dtype))
                p.grad = None
        if dst:
            torch._foreach_add_(dst, src)

    def _piece_done(self, block):
        """called (from a tensor hook) when the gradient has reached the input of `block`: everything behind block + 1 in
        the arena is final -> gather it and hand it to the side stream"""
        lo = self._block_start[block + 1]
        hi = self._reduced_from if self._reduced_from is not None else self.arena.grad.numel()
        if lo >= hi:
            return
        self.bank.join_side()          # the dense layers' dW launches of blocks > `block` run on the side stream
        self._gather(lo, hi)
        self.reducer.all_reduce(self.arena.grad[lo:hi], async_op=self.arena.grad.is_cuda, average=True)
        self._reduced_from = lo
