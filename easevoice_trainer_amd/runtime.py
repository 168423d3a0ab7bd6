"""Device-side runtime of a model: flat fp32 parameter / gradient / optimiser-state arenas sized for one
288 GB HBM3E device, the prepared-weight bank, and the flat AdamW step.

Why flat arenas (MI355X-first, not how the reference does it): the reference holds ~900 separate tensors per
model and runs torch.optim.AdamW's per-tensor loops plus a python grad-norm with one `.item()` host sync per
parameter (commons.py:140-155).  Here one launch updates the whole model, one launch computes the grad norm,
one memset zeroes the gradients, and the data-parallel all-reduce (dist.py) works on a few large contiguous
buckets that keep the 7 xGMI links busy instead of hundreds of small messages.
"""
import ctypes as C

import torch
from torch import nn

from .hip import lib as L
from .hip.conv import WeightBank

_ALIGN = 64  # floats: every parameter starts on a 256-byte boundary


class ParamArena:
    """Moves every parameter of `model` into one contiguous fp32 buffer (views keep their names/shapes, so
    state_dict()/load_state_dict() are unchanged) and gives each a `.grad` view into a matching grad buffer."""

    def __init__(self, model: nn.Module, device, reserve=None):
        """reserve: {parameter name: floats} -- room kept (and left zero) behind a parameter, e.g. for a weight whose
        GEMM image is padded to more rows than the parameter has (hip/linear.py)"""
        self.device = torch.device(device)
        params = [(n, p) for n, p in model.named_parameters()]
        # sub-modules may ask for some of their parameters to sit next to each other (`arena_adjacent()`: lists of
        # relative names), e.g. the q / k / v projection weights of an attention layer, which then form ONE dense
        # [3C, C] matrix for a single GEMM launch (hip/conv.py::PackedConv).  The arena order is otherwise the
        # registration order; nothing but the offsets depends on it.
        order = [n for n, _ in params]
        for mod_name, mod in model.named_modules():
            fn = getattr(mod, "arena_adjacent", None)
            if fn is None:
                continue
            for group in fn():
                full = [f"{mod_name}.{r}" if mod_name else r for r in group]
                if any(n not in order for n in full):
                    continue
                first = min(order.index(n) for n in full)
                pos = sum(1 for n in order[:first] if n not in full)
                rest = [n for n in order if n not in full]
                order = rest[:pos] + full + rest[pos:]
        by_name = dict(params)
        params = [(n, by_name[n]) for n in order]
        offs, total = {}, 0
        reserve = reserve or {}
        for n, p in params:
            offs[n] = total
            total += (max(p.numel(), int(reserve.get(n, 0))) + _ALIGN - 1) // _ALIGN * _ALIGN
        self.numel = total
        self.updates = 0          # bumped by every optimiser launch on this arena (raw-pointer writes torch cannot see)
        self.param = torch.zeros(total, dtype=torch.float32, device=self.device)
        self.grad = torch.zeros(total, dtype=torch.float32, device=self.device)
        self.offsets = offs
        self.names = [n for n, _ in params]
        for n, p in params:
            o = offs[n]
            view = self.param[o:o + p.numel()].view(p.shape)
            view.copy_(p.data.to(self.device, torch.float32))
            p.data = view
            p.grad = self.grad[o:o + p.numel()].view(p.shape)
        self.model = model

    def range_of(self, name, numel):
        o = self.offsets[name]
        return o, o + numel

    def range_of_prefix(self, prefix, stop_before=None):
        """[lo, hi) floats of all parameters whose name starts with `prefix` (registration order keeps a sub-module's
        parameters adjacent), optionally ending before the first parameter that starts with `stop_before`; hi is
        aligned up like every parameter start.  Used to reduce one sub-model's gradients on their own."""
        names = [n for n in self.names if n.startswith(prefix)]
        if stop_before is not None:
            cut = [i for i, n in enumerate(names) if n.startswith(stop_before)]
            if cut:
                names = names[:cut[0]]
        if not names:
            raise KeyError(prefix)
        i0, i1 = self.names.index(names[0]), self.names.index(names[-1])
        if self.names[i0:i1 + 1] != names:
            raise ValueError(f"parameters of {prefix!r} are not adjacent in the arena")
        hi = self.offsets[self.names[i1 + 1]] if i1 + 1 < len(self.names) else self.numel
        return self.offsets[names[0]], hi

    def zero_grad(self):
        self.grad.zero_()


class FlatAdamW:
    """torch.optim.AdamW semantics (src/train/sovits.py:294-319: lr, betas, eps, weight_decay=0.01 default,
    per-group learning rates) over a ParamArena: ONE launch per step.  Parameters listed in `frozen` (no
    gradient in the reference, e.g. ssl_proj: models.py:912-921) are skipped exactly as AdamW skips
    `grad is None` parameters."""

    def __init__(self, arena: ParamArena, groups, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        """groups: list of dict(names=[param names], lr=float[, slots=[param names]]).  `slots` is the group's parameter
        list as the reference's optimizer holds it (it may contain parameters that never receive a gradient and so are
        not in `names`); it only fixes the numbering of state_dict(), so that checkpoints resume in either direction."""
        self.arena, self.betas, self.eps, self.weight_decay = arena, betas, eps, weight_decay
        self.exp_avg = torch.zeros_like(arena.param)
        self.exp_avg_sq = torch.zeros_like(arena.param)
        self.step_count = 0
        self._step_dev = torch.zeros(1, dtype=torch.int32, device=arena.device)   # the kernels' copy of step_count
        self.groups = []
        named = dict(arena.model.named_parameters())
        for g in groups:
            ranges = sorted(arena.range_of(n, named[n].numel()) for n in g["names"])
            merged = []
            for b, e in ranges:
                e_al = (e + _ALIGN - 1) // _ALIGN * _ALIGN
                if merged and merged[-1][1] == b:
                    merged[-1][1] = e_al
                else:
                    merged.append([b, e_al])
            self.groups.append(dict(lr=g["lr"], initial_lr=g["lr"], ranges=merged, names=list(g["names"]),
                                    slots=list(g.get("slots", g["names"]))))
        self._table = None
        self._table_lrs = None

    @property
    def param_groups(self):
        return self.groups

    def _segments(self):
        lrs = tuple(g["lr"] for g in self.groups)
        if self._table is None or lrs != self._table_lrs:
            segs = []
            for g in self.groups:
                for b, e in g["ranges"]:
                    segs.append(L.AdamWSeg(b, e, g["lr"], self.weight_decay))
            if len(segs) > 64:
                raise L.EvtError(f"{len(segs)} AdamW segments (max 64): register parameters group-contiguously")
            new = L.struct_to_device(segs, self.arena.device)
            if self._table is None:
                self._table = new
            else:
                self._table.copy_(new)     # in place: captured graphs keep reading the same device table
            self._nseg = len(segs)
            self._table_lrs = lrs
        return self._table, self._nseg

    def step(self, grad_scale=1.0, skip=None):
        """skip: a device fp32 flag (DeviceGradScaler.found_inf): non-zero -> the launch changes nothing, the device step
        number included (GradScaler.step on an overflow).  The python-side step_count then runs ahead of the device's;
        sync_step_count() re-reads it (checkpoints do)."""
        self.step_count += 1
        tab, nseg = self._segments()
        a = self.arena
        a.updates += 1
        if skip is not None:
            self._guarded = True
            L.check(L.lib().evt_adamw_flat_dev_guarded(L.ptr(a.param), L.ptr(a.grad), L.ptr(self.exp_avg),
                                                       L.ptr(self.exp_avg_sq), C.c_int64(a.numel), L.ptr(tab), nseg,
                                                       C.c_float(self.betas[0]), C.c_float(self.betas[1]), C.c_float(self.eps),
                                                       L.ptr(self._step_dev), C.c_float(grad_scale), L.ptr(skip),
                                                       L.stream_ptr()), "evt_adamw_flat_dev_guarded")
            return
        L.check(L.lib().evt_adamw_flat_dev(L.ptr(a.param), L.ptr(a.grad), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq),
                                           C.c_int64(a.numel), L.ptr(tab), nseg, C.c_float(self.betas[0]),
                                           C.c_float(self.betas[1]), C.c_float(self.eps), L.ptr(self._step_dev),
                                           C.c_float(grad_scale), L.stream_ptr()), "evt_adamw_flat_dev")

    def sync_step_count(self):
        """the number of updates actually applied (the device counter: guarded steps that were skipped do not count)"""
        if getattr(self, "_guarded", False):
            self.step_count = int(self._step_dev.item())
        return self.step_count

    def step_range(self, lo, hi, bump, grad_scale=1.0):
        """the update of the arena elements [lo, hi) only (evt_adamw_flat_dev_range): one sub-model's parameters, as soon as
        its gradients are complete.  The ranges of one step partition the arena; the first one enqueued passes bump=True
        (the step number moves once), the others must be ordered after it on the device."""
        if bump:
            self.step_count += 1
            self.arena.updates += 1
        tab, nseg = self._segments()
        a = self.arena
        L.check(L.lib().evt_adamw_flat_dev_range(L.ptr(a.param), L.ptr(a.grad), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq),
                                                 C.c_int64(lo), C.c_int64(hi), L.ptr(tab), nseg, C.c_float(self.betas[0]),
                                                 C.c_float(self.betas[1]), C.c_float(self.eps), L.ptr(self._step_dev),
                                                 1 if bump else 0, C.c_float(grad_scale), L.stream_ptr()),
                "evt_adamw_flat_dev_range")

    def note_replayed_step(self):
        """a captured graph ran the update: keep the python-side counter (checkpoints) in step with the device one"""
        self.step_count += 1
        self.arena.updates += 1

    # ---- torch.optim-compatible checkpoint surface (src/utils/path/ckpt.py:78-93 stores optimizer.state_dict()) ----
    def state_dict(self):
        """torch.optim.AdamW.state_dict() layout: parameters numbered group by group in `slots` order; entries only for
        parameters that are updated (a parameter without gradients has no state in torch either)"""
        self.sync_step_count()
        named = dict(self.arena.model.named_parameters())
        state, idx, pgroups = {}, 0, []
        for g in self.groups:
            ids, live = [], set(g["names"])
            for n in g["slots"]:
                if n in live:
                    b, e = self.arena.range_of(n, named[n].numel())
                    state[idx] = dict(step=torch.tensor(float(self.step_count)),
                                      exp_avg=self.exp_avg[b:e].view(named[n].shape).clone(),
                                      exp_avg_sq=self.exp_avg_sq[b:e].view(named[n].shape).clone())
                ids.append(idx)
                idx += 1
            pgroups.append(dict(lr=g["lr"], initial_lr=g["initial_lr"], betas=self.betas, eps=self.eps,
                                weight_decay=self.weight_decay, amsgrad=False, params=ids))
        return dict(state=state, param_groups=pgroups)

    def load_state_dict(self, sd):
        named = dict(self.arena.model.named_parameters())
        idx = 0
        for g, pg in zip(self.groups, sd["param_groups"]):
            if len(pg["params"]) != len(g["slots"]):
                raise ValueError(f"optimizer state has {len(pg['params'])} parameters in a group of {len(g['slots'])}")
            g["lr"] = pg["lr"]
            g["initial_lr"] = pg.get("initial_lr", pg["lr"])
            live = set(g["names"])
            for n, pid in zip(g["slots"], pg["params"]):
                st = sd["state"].get(pid)
                if st is not None and n in live:
                    b, e = self.arena.range_of(n, named[n].numel())
                    self.exp_avg[b:e].copy_(st["exp_avg"].reshape(-1))
                    self.exp_avg_sq[b:e].copy_(st["exp_avg_sq"].reshape(-1))
                    self.step_count = max(self.step_count, int(float(st["step"])))
                idx += 1
        self._step_dev.fill_(self.step_count)


class ModelRuntime:
    """arena + weight bank for one nn.Module; `prepare()` before a forward, `finish_grads()` after backward."""

    def __init__(self, model: nn.Module, dtype=torch.float32, device="cuda:0", impl=L.IMPL_AUTO):
        L.lib()  # fail loudly before touching the device if the extension is missing
        self.model, self.dtype, self.device = model, dtype, torch.device(device)
        model.to(self.device)
        self.arena = ParamArena(model, self.device)
        self.bank = WeightBank(model, dtype, self.device, impl=impl)
        self.bank.build_tables()
        for m in model.modules():
            if hasattr(m, "cd"):
                m.cd = dtype
        self._sumsq = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._fold_stamp = None
        self._book = None
        # Parameters whose gradient comes from autograd (linears, norms, embeddings -- everything that is not a fused
        # conv): autograd's AccumulateGrad would run one `grad += new` launch per parameter per backward (~420 tiny
        # launches in the s2 generator).  Instead their .grad is detached from the arena during the backward (autograd
        # then just keeps the produced tensor) and finish_grads() gathers all of them with one multi-tensor copy.
        conv_owned = set()
        for m in model.modules():
            if hasattr(m, "_slot"):
                conv_owned.update(id(p) for p in m.parameters(recurse=False))
        self._free = [(p, p.grad) for p in model.parameters() if id(p) not in conv_owned]
        # Kernels that produce such a parameter's gradient themselves (LayerNorm gamma / beta, relative-position
        # embeddings: fp32 atomics) accumulate straight into the arena view and hand autograd no tensor at all
        # (hip/enc.py::grad_sink) -- no zero-filled scratch, no clone in AccumulateGrad, nothing to gather afterwards.
        for p, view in self._free:
            p._evt_grad_view = view

    def zero_grad(self):
        self.arena.zero_grad()
        self.bank.zero_dw()
        for p, _view in self._free:
            p.grad = None

    def prepare(self, force: bool = False):
        """fold the prepared conv weight images -- only when the parameters changed since the last fold: an optimiser
        step of this arena (FlatAdamW bumps `arena.updates`) or an in-place write that torch versions, through the arena
        (broadcast) or through a parameter (load_state_dict, p.copy_).  The discriminator is refolded right after its update for the generator step, so the fold at the
        top of the next step would rebuild identical images."""
        a = self.arena
        # each Parameter carries its own version counter (p.data = view): load_state_dict / p.copy_ bump the parameter's,
        # not the arena's, so both are in the stamp
        stamp = (a.updates, a.param._version, sum(p._version for p in self.model.parameters()))
        if force or stamp != self._fold_stamp:
            self.bank.fold()
            self._fold_stamp = stamp

    def finish_conv_grads(self, modules):
        """the convolution weight gradients of `modules` only (a contiguous run of the bank), e.g. one sub-discriminator
        right after its backward; the final finish_grads(done=[...]) must then be told which row ranges are done"""
        lo, hi = self.bank.rows_of(modules)
        self.bank.grads(lo, hi)
        return lo, hi

    def finish_grads(self, done=()):
        """`done`: row ranges already finished by finish_conv_grads() in this backward (each row exactly once)"""
        if not done:
            self.bank.grads()
        else:
            at = 0
            for lo, hi in sorted(done):
                self.bank.grads(at, lo)
                at = hi
            self.bank.grads(at, None)
        self.gather_free_grads()

    # ---- pipelined bookkeeping: a sub-model's weight-norm gradient, AdamW update and refold run on a side stream as
    #      soon as its backward is done, under the backward of the next sub-model.  These three launches are HBM streams
    #      over the parameters (2.0 of the 24.5 ms s2 step when they run after the backward, on the main stream); the
    #      backward next to them is MFMA / latency bound.  Same kernels, same per-element arithmetic as the serial order.
    def book_stream(self):
        if self._book is None:
            self._book = L.role_stream(self.device, "book")
        return self._book

    def gather_free_grads(self):
        """autograd-owned gradients -> their arena views (what finish_grads does after the convolutions' rows)"""
        dst, src = [], []
        for p, view in self._free:
            if p.grad is not None and p.grad.data_ptr() != view.data_ptr():
                dst.append(view)
                src.append(p.grad if p.grad.dtype == view.dtype else p.grad.to(view.dtype))
            p.grad = view
        if dst:
            torch._foreach_copy_(dst, src)

    def book_piece(self, optim, grad_rows, fold_rows, ranges, first, grad_scale=1.0):
        """grad_rows / fold_rows: [(lo, hi)] of the bank's row table (rows_of); ranges: [(lo, hi)] floats of the arena.
        Enqueued on the bookkeeping stream behind everything the current stream holds so far and behind the queued weight
        gradients.  A convolution whose parameters are updated by a LATER piece must not be in fold_rows here."""
        bank = self.bank
        bank.flush_deferred()
        bank.check_tables()
        cur, book = torch.cuda.current_stream(self.device), self.book_stream()
        book.wait_stream(cur)
        if bank._side is not None:
            book.wait_stream(bank._side)
        with torch.cuda.stream(book):
            for lo, hi in grad_rows:
                bank.grads(lo, hi, join=False)
            n = 0
            for lo, hi in ranges:
                if hi > lo:
                    optim.step_range(lo, hi, first and n == 0, grad_scale)
                    n += 1
            for lo, hi in fold_rows:
                bank.fold(lo, hi)

    def book_join(self, all_folded=True):
        """the current stream waits for the bookkeeping stream; with all_folded the images are marked current (the
        prepare() at the top of the next step then has nothing to do)"""
        if self._book is not None:
            torch.cuda.current_stream(self.device).wait_stream(self._book)
        self.bank.join_side()
        if all_folded:
            self.mark_folded()

    def mark_folded(self):
        a = self.arena
        self._fold_stamp = (a.updates, a.param._version, sum(p._version for p in self.model.parameters()))

    def grad_sumsq(self):
        """sum of squares of all gradients, as a device scalar (no host sync)"""
        self._sumsq.zero_()
        L.check(L.lib().evt_sumsq(L.ptr(self.arena.grad), C.c_int64(self.arena.numel), L.ptr(self._sumsq),
                                  L.stream_ptr()), "evt_sumsq")
        return self._sumsq


class DeviceGradScaler:
    """torch.cuda.amp.GradScaler (src/train/sovits.py:378: `GradScaler(enabled=hps.train.fp16_run)`; :504-507 and :521-525:
    scale(loss).backward(), unscale_(optim), step(optim), update()) with its state in device memory: the scale, the growth
    tracker and one found-inf flag per optimiser are tensors the kernels read and write, so a step has no host
    synchronisation (GradScaler.step reads found_inf on the host: `.item()`) and is captured into the HIP graphs like the
    rest.  Same arithmetic as torch's: _amp_foreach_non_finite_check_and_unscale_ (evt_scaler_unscale), the skipped
    optimiser step (evt_adamw_flat_dev_guarded), _amp_update_scale_ (evt_scaler_update).  One scaler serves both
    optimisers of the GAN step, as in the reference: update() looks at both flags."""

    def __init__(self, device, init_scale=2.0 ** 16, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000,
                 enabled=True):
        self.device = torch.device(device)
        self.enabled = enabled
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval
        self._scale = torch.full((1,), float(init_scale), dtype=torch.float32, device=self.device)
        self._tracker = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._found = {}            # id(optimiser) -> fp32 flag [1]
        self._flag_table = None

    def found_inf(self, owner):
        """the flag of one optimiser; `owner` is anything that identifies it (the engines pass the model's runtime)"""
        f = self._found.get(id(owner))
        if f is None:
            f = self._found[id(owner)] = torch.zeros(1, dtype=torch.float32, device=self.device)
            self._flag_table = None
        return f

    def scale(self, loss):
        """loss * scale as a device-side product (the seed of the backward carries the scale)"""
        return loss * self._scale[0] if self.enabled else loss

    def unscale_(self, owner):
        """owner.arena.grad *= 1 / scale; owner's flag is raised if a gradient is not finite.  The flag is cleared by
        update() only (torch.amp.GradScaler keeps found_inf per optimizer until update() as well): a step that unscales
        must end with update(), as both trainers' loops do -- a do_opt=False caller that skips it inherits the flag."""
        if not self.enabled:
            return
        a = owner.arena
        L.check(L.lib().evt_scaler_unscale(L.ptr(a.grad), C.c_int64(a.numel), L.ptr(self._scale), L.ptr(self.found_inf(owner)),
                                           L.stream_ptr()), "evt_scaler_unscale")

    def update(self):
        if not self.enabled or not self._found:
            return
        if self._flag_table is None:
            flags = list(self._found.values())
            self._flag_table = (C.c_void_p * len(flags))(*[f.data_ptr() for f in flags])
        L.check(L.lib().evt_scaler_update(L.ptr(self._scale), L.ptr(self._tracker), self._flag_table, len(self._found),
                                          C.c_float(self.growth_factor), C.c_float(self.backoff_factor),
                                          int(self.growth_interval), L.stream_ptr()), "evt_scaler_update")

    def get_scale(self):
        return float(self._scale.item()) if self.enabled else 1.0

    def state_dict(self):
        """torch.amp.GradScaler.state_dict() layout"""
        if not self.enabled:
            return {}
        return {"scale": self.get_scale(), "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "_growth_tracker": int(self._tracker.item())}

    def load_state_dict(self, sd):
        if not self.enabled or not sd:
            return
        self._scale.fill_(float(sd["scale"]))
        self._tracker.fill_(int(sd["_growth_tracker"]))
        self.growth_factor, self.backoff_factor = sd["growth_factor"], sd["backoff_factor"]
        self.growth_interval = sd["growth_interval"]
