"""ScaledAdam (Povey) over a flat parameter arena, and the reference's LR schedule.

Arithmetic of src/easevoice/soundstorm/auto_reg/modules/optim.py:206-251 (step), :300-390 (_get_clipping_scale),
:448-497 (_step_one_batch), :499-558 (_size_update), :560-598 (_step), :600-622 (_step_scalar), restated per tensor
(the reference's same-shape stacking is an implementation detail of its torch ops, not of the maths): two HIP launches
over the whole model per optimiser step plus O(#tensors) vector ops, instead of 11 stacked batches of torch ops with
stack/unstack copies of every parameter and gradient."""
import ctypes as C

import torch

from ..hip import lib as L

_CHUNK = 1 << 16


class ScaledAdam:
    def __init__(self, arena, lr=3e-2, clipping_scale=None, betas=(0.9, 0.98), scalar_lr_scale=0.1, eps=1e-8,
                 param_min_rms=1e-5, param_max_rms=3.0, scalar_max=10.0, size_update_period=4,
                 clipping_update_period=100, exclude=()):
        self.arena, self.lr, self.clipping_scale, self.betas = arena, lr, clipping_scale, betas
        self.scalar_lr_scale, self.eps, self.param_min_rms, self.param_max_rms = scalar_lr_scale, eps, param_min_rms, param_max_rms
        self.scalar_max, self.P, self.clip_period = scalar_max, size_update_period, clipping_update_period
        dev = arena.device
        named = [(n, p) for n, p in arena.model.named_parameters() if n not in exclude and p.requires_grad]
        self.names = [n for n, _ in named]
        chunks, numel = [], []
        for t, (n, p) in enumerate(named):
            b, e = arena.range_of(n, p.numel())
            numel.append(p.numel())
            for c0 in range(b, e, _CHUNK):
                chunks.append(L.SAChunk(c0, min(e, c0 + _CHUNK), t, 0))
        self.nt, self.nchunks = len(named), len(chunks)
        self._chunk_list = [(c.begin, c.end, c.tensor) for c in chunks]
        self._chunks = L.struct_to_device(chunks, dev)
        self.numel = torch.tensor(numel, dtype=torch.float32, device=dev)
        self.is_scalar = (self.numel == 1).float()
        self.delta = torch.zeros_like(arena.param)
        self.exp_avg_sq = torch.zeros_like(arena.param)
        self.param_rms = torch.zeros(self.nt, device=dev)
        self.scale_exp_avg_sq = torch.zeros(self.nt, device=dev)
        self.scale_grads = torch.zeros(self.P, self.nt, device=dev)
        self.model_norms = torch.zeros(self.clip_period, device=dev)
        self.model_norm_threshold = None
        self.num_clipped = 0
        self.step_count = 0
        self._stats = torch.zeros(self.nt, 3, device=dev)
        self._coef = torch.zeros(self.nt, 4, device=dev)
        self._clip = torch.ones(1, device=dev)
        self.param_groups = [dict(lr=lr)]

    def _clipping(self, gg):
        """optim.py:300-390; returns a device scalar tensor"""
        step = self.step_count
        if self.clipping_scale is None or step == 0:
            return torch.ones(1, device=gg.device)
        tot_sumsq = torch.where(self.is_scalar > 0, gg, gg * self.param_rms ** 2).sum()
        tot_norm = tot_sumsq.sqrt()
        self.model_norms[step % self.clip_period] = tot_norm
        if step % self.clip_period == 0:
            sorted_norms = self.model_norms.sort()[0]
            median = sorted_norms[min(self.clip_period - 1, (self.clip_period // 4) * 2)]
            self.model_norm_threshold = self.clipping_scale * median      # stays on device
        if step < self.clip_period or self.model_norm_threshold is None:
            return torch.ones(1, device=gg.device)
        return torch.clamp(self.model_norm_threshold / (tot_norm + 1.0e-20), max=1.0).reshape(1)

    def _k_stats(self):
        a = self.arena
        L.check(L.lib().evt_scaled_adam_stats(L.ptr(a.param), L.ptr(a.grad), L.ptr(self._chunks), self.nchunks,
                                              L.ptr(self._stats), L.stream_ptr()), "evt_scaled_adam_stats")

    def _k_apply(self, hp):
        a = self.arena
        L.check(L.lib().evt_scaled_adam_apply(L.ptr(a.param), L.ptr(a.grad), L.ptr(self.delta), L.ptr(self.exp_avg_sq),
                                              L.ptr(self._chunks), self.nchunks, L.ptr(self._coef), C.byref(hp),
                                              L.stream_ptr()), "evt_scaled_adam_apply")

    @torch.no_grad()
    def step(self):
        a = self.arena
        lr = self.param_groups[0]["lr"]
        beta1, beta2 = self.betas
        step, P = self.step_count, self.P
        self._stats.zero_()
        self._k_stats()
        pg, pp, gg = self._stats[:, 0], self._stats[:, 1], self._stats[:, 2]
        if step == 0:   # _init_state (optim.py:253-298): param_rms from the initial parameters
            self.param_rms.copy_((pp / self.numel).sqrt())
        clip = self._clipping(gg)
        self.scale_grads[step % P] = pg * clip
        coef0 = torch.zeros(self.nt, device=a.device)
        if step % P == P - 1:
            self.param_rms.copy_((pp / self.numel).sqrt())
            if step > 0:
                size_lr = lr * self.scalar_lr_scale
                beta2_corr = beta2 ** P
                self.scale_exp_avg_sq.mul_(beta2_corr).add_((self.scale_grads ** 2).mean(dim=0), alpha=1 - beta2_corr)
                size_step = (step + 1) // P
                bc2 = 1 - beta2_corr ** size_step
                denom = self.scale_exp_avg_sq.sqrt() + self.eps
                scale_step = -size_lr * (bc2 ** 0.5) * self.scale_grads.sum(dim=0) / denom
                scale_step = torch.where(self.param_rms < self.param_min_rms, torch.zeros_like(scale_step), scale_step)
                scale_step = torch.where(self.param_rms > self.param_max_rms,
                                         torch.full_like(scale_step, -size_lr * P), scale_step)
                coef0 = scale_step * (1 - beta1) * (1 - self.is_scalar)
        alpha = -lr * (1 - beta1) * self.param_rms.clamp(min=self.param_min_rms)
        self._coef[:, 0], self._coef[:, 1], self._coef[:, 2] = coef0, alpha, self.is_scalar
        hp = L.ScaledAdamHP(lr, beta1, beta2, self.eps, self.scalar_lr_scale, self.scalar_max, step, 0)
        self._k_apply(hp)
        self.step_count += 1

    def zero_grad(self):
        self.arena.zero_grad()

    def state_dict(self):
        return dict(step=self.step_count, delta=self.delta, exp_avg_sq=self.exp_avg_sq, param_rms=self.param_rms,
                    scale_exp_avg_sq=self.scale_exp_avg_sq, scale_grads=self.scale_grads, model_norms=self.model_norms,
                    model_norm_threshold=self.model_norm_threshold, names=self.names, lr=self.param_groups[0]["lr"])

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        for k in ("delta", "exp_avg_sq", "param_rms", "scale_exp_avg_sq", "scale_grads", "model_norms"):
            getattr(self, k).copy_(sd[k])
        self.model_norm_threshold = sd.get("model_norm_threshold")
        self.param_groups[0]["lr"] = sd.get("lr", self.lr)


    # ---- the reference optimiser's own state layout (what a Lightning checkpoint of src/train/gpt.py holds under
    #      "optimizer_states"[0]): torch.optim.Optimizer.state_dict() of its BatchedOptimizer -- parameters of equal
    #      (dtype, shape) are stacked, batches sorted by that key, the state of a batch stored under its FIRST parameter's
    #      index with a leading stacking dimension (optim.py:60-127, :253-298); the first batch also carries the clipping
    #      history (optim.py:339-362).  `order`: parameter names in the reference's param_groups order (= the order of the
    #      checkpoint's state_dict keys). ----
    def _ref_batches(self, order):
        named = dict(self.arena.model.named_parameters())
        mine = {n: t for t, n in enumerate(self.names)}
        groups = {}
        for idx, n in enumerate(order):
            if n not in mine:
                raise KeyError(f"optimizer state: parameter {n!r} of the checkpoint is not optimised here")
            p = named[n]
            groups.setdefault(("torch.float32", *p.shape), []).append((idx, n, mine[n], p))
        if len(groups) == 0 or sum(len(v) for v in groups.values()) != len(self.names):
            raise ValueError("optimizer state: the checkpoint's parameter list differs from this model's")
        return [groups[k] for k in sorted(groups)]

    @torch.no_grad()
    def load_reference_state(self, sd, order):
        """continue a run the reference started: `sd` = its optimizer.state_dict()"""
        state = sd["state"]
        a = self.arena
        for bi, batch in enumerate(self._ref_batches(order)):
            st = state[batch[0][0]]
            self.step_count = int(st["step"])
            for i, (_idx, n, t, p) in enumerate(batch):
                b, e = a.range_of(n, p.numel())
                self.delta[b:e].copy_(st["delta"][i].reshape(-1))
                self.exp_avg_sq[b:e].copy_(st["exp_avg_sq"][i].reshape(-1))
                if p.numel() > 1:
                    self.param_rms[t] = st["param_rms"][i].reshape(())
                    self.scale_exp_avg_sq[t] = st["scale_exp_avg_sq"][i].reshape(())
                    self.scale_grads[:, t] = st["scale_grads"][:, i].reshape(self.P)
            if bi == 0:
                if "model_norms" in st:
                    self.model_norms.copy_(st["model_norms"])
                thr = st.get("model_norm_threshold")
                self.model_norm_threshold = None if thr is None else torch.as_tensor(float(thr), device=a.device)
                self.num_clipped = int(st.get("num_clipped", 0))
        self.param_groups[0]["lr"] = sd["param_groups"][0].get("lr", self.lr)

    @torch.no_grad()
    def reference_state_dict(self, order):
        """this optimiser's state in the reference optimiser's layout (its load_state_dict takes it)"""
        a = self.arena
        state = {}
        for bi, batch in enumerate(self._ref_batches(order)):
            shape = batch[0][3].shape
            cut = lambda buf: torch.stack([buf[slice(*a.range_of(n, p.numel()))].reshape(shape).cpu().clone()
                                           for _i, n, _t, p in batch])
            st = dict(step=self.step_count, delta=cut(self.delta), exp_avg_sq=cut(self.exp_avg_sq))
            ts = [t for _i, _n, t, _p in batch]
            stacked_numel = len(batch) * batch[0][3].numel()          # optim.py:279-281: the STACKED tensor decides
            if stacked_numel > 1:
                one = (len(batch),) + (1,) * len(shape)
                st["param_rms"] = self.param_rms[ts].reshape(one).cpu().clone()
                st["scale_exp_avg_sq"] = self.scale_exp_avg_sq[ts].reshape(one).cpu().clone()
                st["scale_grads"] = self.scale_grads[:, ts].reshape((self.P,) + one).cpu().clone()
            if bi == 0 and self.clipping_scale is not None and self.step_count > 0:
                st["model_norms"] = self.model_norms.cpu().clone()
                if self.model_norm_threshold is not None:
                    st["model_norm_threshold"] = float(self.model_norm_threshold)
                st["num_clipped"] = int(self.num_clipped)
            state[batch[0][0]] = st
        beta1, beta2 = self.betas
        group = dict(lr=self.param_groups[0]["lr"], clipping_scale=self.clipping_scale, betas=(beta1, beta2),
                     scalar_lr_scale=self.scalar_lr_scale, eps=self.eps, param_min_rms=self.param_min_rms,
                     param_max_rms=self.param_max_rms, scalar_max=self.scalar_max, size_update_period=self.P,
                     clipping_update_period=self.clip_period, params=list(range(len(order))))
        return dict(state=state, param_groups=[group])


class WarmupCosineLRSchedule:
    """src/easevoice/soundstorm/auto_reg/modules/lr_schedulers.py:11-65.  The reference computes the warm-up / cosine
    value and then OVERWRITES it: every step() pins the optimiser's lr to 0.002 (line 61).  Reproduced as is."""

    def __init__(self, optimizer, init_lr, peak_lr, end_lr, warmup_steps=10000, total_steps=400000, current_step=0):
        self.init_lr, self.peak_lr, self.end_lr, self.optimizer = init_lr, peak_lr, end_lr, optimizer
        self.warmup_steps, self.total_steps, self._current_step = warmup_steps, total_steps, current_step
        # the two slopes are fixed at construction (from the end_lr given here: step() overwrites end_lr afterwards)
        self._warmup_rate = (peak_lr - init_lr) / warmup_steps
        self._decay_rate = (end_lr - peak_lr) / (total_steps - warmup_steps)
        self.lr = init_lr
        self._last_lr = [self.lr]

    def get_last_lr(self):
        return self._last_lr

    def step(self):
        self.lr = self.end_lr = 0.002
        self._last_lr = [g["lr"] for g in self.optimizer.param_groups]
        for g in self.optimizer.param_groups:
            g["lr"] = self.end_lr
        self._current_step += 1
        return self.lr

    def state_dict(self):
        """torch's _LRScheduler.state_dict(): every attribute but the optimiser (what a Lightning checkpoint stores under
        "lr_schedulers")"""
        return {k: v for k, v in vars(self).items() if k != "optimizer"}

    def load_state_dict(self, sd):
        for k in ("init_lr", "peak_lr", "end_lr", "warmup_steps", "total_steps", "_current_step", "lr", "_last_lr",
                  "_warmup_rate", "_decay_rate"):
            if k in sd:
                setattr(self, k, sd[k])
