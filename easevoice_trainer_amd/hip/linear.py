"""Host side of the dense-layer GEMMs (C ABI: evt_gemm_bf16_*, csrc/gemm.hip): a bank of prepared weight images for
all Linear layers of a model and the autograd Function that calls the three entry points.

Reference modules mirrored: torch.nn.Linear / the packed `in_proj_weight` of MultiheadAttention, called through
F.linear at src/easevoice/soundstorm/auto_reg/modules/transformer.py:207-224,330-334,
patched_mha_with_cache.py:242,460 and models/t2s_model.py:276,486.  The parameters stay what they are in the
reference (fp32 `weight` [N, K] / `bias` [N] with the same state_dict keys); what is added is a bf16 (or fp32) image of
W and of W^T per layer, rebuilt by ONE multi-tensor launch whenever the optimiser has touched the weights -- instead of
the per-call `weight.to(bf16)` casts of a torch-level implementation.
"""
import ctypes as C
import os

import torch

from . import lib as L

N_ALIGN = 128      # an output width that is not a multiple of 128 is padded up to it (zero rows / columns)


class GemmParams(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("relu", C.c_int32)]


class GemmEpilogue(C.Structure):
    """evt_gemm_epilogue (include/evt.h)"""
    _fields_ = [("dropout_p", C.c_float), ("site", C.c_uint32), ("seed_dev", C.c_void_p), ("gate", C.c_void_p),
                ("gate_pos", C.c_float), ("pad_", C.c_uint32), ("add", C.c_void_p)]


class LinearSlot:
    __slots__ = ("name", "weight", "bias", "N", "Np", "K", "layout", "reg", "alt", "bank", "_pcache")

    def __init__(self, name, weight, bias, bank):
        self.name, self.weight, self.bias, self.bank = name, weight, bias, bank
        self.N, self.K = weight.shape
        self.Np = (self.N + N_ALIGN - 1) // N_ALIGN * N_ALIGN if self.N % 8 else self.N
        if self.Np != self.N and bias is not None:
            raise L.EvtError(f"linear {name}: a padded output width ({self.N} -> {self.Np}) with a bias is not supported")
        self._pcache = {}

    def params(self, M, relu):
        p = self._pcache.get((M, relu))
        if p is None:
            p = self._pcache[(M, relu)] = GemmParams(self.bank.dt, M, self.Np, self.K, 1 if relu else 0)
        return p

    def fused(self, M, backward_data):
        """does the 256 x 256 kernel with fused epilogues (csrc/gemm256.hip) cover this layer at M rows?"""
        key = ("fused", M, backward_data)
        f = self._pcache.get(key)
        if f is None:
            f = self._pcache[key] = bool(L.lib().evt_gemm_bf16_fused_supported(C.byref(self.params(M, False)),
                                                                               1 if backward_data else 0))
        return f


class LinearBank:
    """Prepared weight images of a set of Linear weights.  `specs`: iterable of (name, weight Parameter [N, K], bias
    Parameter [N] or None).  A weight whose N is padded (slot.Np > N) must sit in storage that is readable -- and zero
    -- up to Np rows (runtime.ParamArena(reserve=...))."""

    def __init__(self, specs, dtype: torch.dtype, device):
        L.set_half(dtype)
        self.dt = L.dt_code(dtype)
        self.dtype, self.device = dtype, torch.device(device)
        self.anchor = torch.zeros(1, device=device, requires_grad=True)
        self.slots = []
        reg_n = alt_n = 0
        offs = []
        for name, w, b in specs:
            s = LinearSlot(name, w, b, self)
            lay = L.WLayout()
            g = GemmParams(self.dt, 128, s.Np, s.K, 0)
            L.check(L.lib().evt_gemm_bf16_layout(C.byref(g), C.byref(lay)), "evt_gemm_bf16_layout")
            s.layout = lay
            self.slots.append(s)
            offs.append((reg_n, alt_n))
            reg_n += (lay.reg_elems + 127) // 128 * 128
            alt_n += (lay.alt_elems + 127) // 128 * 128
            w._evt_slot = s
        self.reg_arena = torch.zeros(max(reg_n, 1), dtype=dtype, device=device)
        self.alt_arena = torch.zeros(max(alt_n, 1), dtype=dtype, device=device)
        for s, (ro, ao) in zip(self.slots, offs):
            s.reg = self.reg_arena[ro: ro + s.layout.reg_elems]
            s.alt = self.alt_arena[ao: ao + s.layout.alt_elems]
        self._items = self._rows = None
        self._stamp = None
        self.dirty = True
        # weight-gradient side stream (round 6): the dW launches of the dense layers produce parameter gradients only --
        # nothing on the backward's critical path reads them -- so an engine may let them run next to the chain of
        # backward-data GEMMs / attention / LayerNorm launches: their tails (a partial second round of blocks, the
        # memory-side atomics of the split-K epilogue) fill with the next kernel's blocks.  Off until an engine turns it on.
        self.side = None
        self.side_on = False          # only between an engine's side_begin() and join_side(): a caller that differentiates
        self._side_used = False       # through the modules directly gets its weight gradients in stream order
        # ... and WHEN on the side stream: queued, and handed over in one go right before a layer's attention backward
        # (auto_reg/blocks.py) -- four dW GEMMs (MFMA-bound) beside the three attention kernels (VALU-bound, the matrix pipe
        # ~85 % idle) share the CUs better than dW GEMMs beside backward-data GEMMs do.  EVT_S1_WGRAD_PAIR=0: hand over at once.
        self.pair = os.environ.get("EVT_S1_WGRAD_PAIR", "1") != "0"
        self._pending = []

    def enable_side_stream(self):
        if self.device.type == "cuda" and self.side is None:
            self.side = L.role_stream(self.device, "lin", ring=2)

    def side_begin(self):
        """from here to join_side() the weight-gradient launches go to the side stream"""
        self.side_on = self.side is not None

    def flush_pending(self):
        """hand the queued weight-gradient launches to the side stream (behind everything the current stream has enqueued)"""
        if not self._pending:
            return
        self.side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.side):
            for p, x, dy, dw_buf, db_buf in self._pending:
                L.check(L.lib().evt_gemm_bf16_bwd_weight(C.byref(p), L.ptr(x), L.ptr(dy), L.ptr(dw_buf), L.ptr(db_buf),
                                                         L.stream_ptr()), "evt_gemm_bf16_bwd_weight")
        for _p, x, dy, _w, _b in self._pending:
            x.record_stream(self.side)
            dy.record_stream(self.side)
        self._pending.clear()

    def join_side(self):
        """the caller's stream waits for the weight-gradient launches handed to the side stream since the last join"""
        self.flush_pending()
        if self.side is not None and self._side_used:
            torch.cuda.current_stream(self.device).wait_stream(self.side)
            self._side_used = False

    def side_end(self):
        self.join_side()
        self.side_on = False

    def _tables(self):
        items, rows, groups = [], [], []
        for i, s in enumerate(self.slots):
            it = L.WPrepItem()
            it.v, it.g = s.weight.data_ptr(), None
            it.reg, it.alt = s.reg.data_ptr(), s.alt.data_ptr()
            it.dw = it.dv = it.dg = None
            it.lay, it.dtype = s.layout, self.dt
            items.append(it)
            rows.extend((i, r) for r in range(s.layout.d0))
            groups.extend((i, r, min(8, s.layout.d0 - r)) for r in range(0, s.layout.d0, 8))
        self._groups = torch.tensor(groups, dtype=torch.int32, device=self.device).contiguous()
        self._ngroups = len(groups)
        self._items = L.struct_to_device(items, self.device)
        self._rows = torch.tensor(rows, dtype=torch.int32, device=self.device).contiguous()
        self._nrows = len(rows)
        self._ptrs = [s.weight.data_ptr() for s in self.slots]

    def mark_dirty(self):
        """the optimiser wrote the weights through raw pointers"""
        self.dirty = True

    def prepare(self, force=False):
        """rebuild the images when the weights changed: after an optimiser step (mark_dirty) or an in-place write torch
        versions (load_state_dict, copy_)"""
        stamp = sum(s.weight._version for s in self.slots)
        if self._items is None or self._ptrs != [s.weight.data_ptr() for s in self.slots]:
            self._tables()
            force = True
        if force or self.dirty or stamp != self._stamp:
            # groups of eight rows: both images leave as 16-byte pieces (csrc/elementwise.hip: wn_fold8_kernel)
            L.check(L.lib().evt_wn_fold_groups(L.ptr(self._items), L.ptr(self._groups), self._ngroups, L.stream_ptr()),
                    "evt_wn_fold_groups")
            self._stamp, self.dirty = stamp, False


def _rng(device):
    from .enc import rng_counter
    return rng_counter(device)


def gemm_fwd(slot, x, relu=False, drop=None, add=None):
    """y = [dropout](act(x W^T + b)) [+ add] over the last dimension.  drop = (p, site): the inner dropout of the FFN
    (transformer.py:330-334).  On the 256 x 256 kernel the whole epilogue rides in the GEMM's store; other shapes run the
    plain GEMM and the element-wise launches."""
    if x.dtype != slot.bank.dtype or not x.is_contiguous() or x.size(-1) != slot.K:
        raise L.EvtError(f"linear {slot.name}: contiguous [..., {slot.K}] {slot.bank.dtype} expected, got "
                         f"{tuple(x.shape)} {x.dtype} contiguous={x.is_contiguous()}")
    M = x.numel() // slot.K
    y = torch.empty(x.shape[:-1] + (slot.Np,), dtype=x.dtype, device=x.device)
    bias = slot.bias.data if slot.bias is not None else None
    lib = L.lib()
    p_drop, site = drop if drop is not None else (0.0, 0)
    if slot.fused(M, False):
        epi = None
        if p_drop > 0.0 or add is not None:
            epi = GemmEpilogue(float(p_drop), int(site), _rng(x.device).data_ptr(), None, 1.0, 0,
                               add.data_ptr() if add is not None else None)
        L.check(lib.evt_gemm_bf16_fwd_ex(C.byref(slot.params(M, relu)), L.ptr(x), L.ptr(slot.reg), L.ptr(slot.alt),
                                         L.ptr(bias), C.byref(epi) if epi is not None else None, L.ptr(y),
                                         L.stream_ptr()), "evt_gemm_bf16_fwd_ex")
        return y
    L.check(lib.evt_gemm_bf16_fwd(C.byref(slot.params(M, relu and p_drop == 0.0)), L.ptr(x), L.ptr(slot.reg),
                                  L.ptr(slot.alt), L.ptr(bias), L.ptr(y), L.stream_ptr()), "evt_gemm_bf16_fwd")
    if p_drop > 0.0:
        if not relu:
            raise L.EvtError("dropout epilogue without relu is not used by the s1 blocks")
        z = y
        y = torch.empty_like(z)
        L.check(lib.evt_relu_dropout_fwd(L.dt_of(z), L.ptr(z), C.c_float(p_drop), L.ptr(_rng(x.device)), C.c_uint32(site),
                                         None, 0, 0, L.ptr(y), C.c_int64(z.numel()), L.stream_ptr()),
                "evt_relu_dropout_fwd")
    if add is not None:
        y.add_(add)
    return y


def gemm_bwd_data(slot, dy, gate=None, gate_pos=1.0, add=None):
    """dx = (dy W) [* (gate > 0 ? gate_pos : 0)] [+ add];  dy [..., Np] contiguous -> dx [..., K]"""
    M = dy.numel() // slot.Np
    dx = torch.empty(dy.shape[:-1] + (slot.K,), dtype=dy.dtype, device=dy.device)
    lib = L.lib()
    p = slot.params(M, False)
    if slot.fused(M, True):
        epi = None
        if gate is not None or add is not None:
            epi = GemmEpilogue(0.0, 0, None, gate.data_ptr() if gate is not None else None, float(gate_pos), 0,
                               add.data_ptr() if add is not None else None)
        L.check(lib.evt_gemm_bf16_bwd_data_ex(C.byref(p), L.ptr(dy), L.ptr(slot.reg), L.ptr(slot.alt),
                                              C.byref(epi) if epi is not None else None, L.ptr(dx), L.stream_ptr()),
                "evt_gemm_bf16_bwd_data_ex")
        return dx
    L.check(lib.evt_gemm_bf16_bwd_data(C.byref(p), L.ptr(dy), L.ptr(slot.reg), L.ptr(slot.alt), L.ptr(dx),
                                       L.stream_ptr()), "evt_gemm_bf16_bwd_data")
    if gate is not None:     # the same derivative as its own launch (odd shapes, the fp32 parity path)
        g = torch.empty_like(dx)
        L.check(lib.evt_dact_mul(L.dt_of(dx), L.ptr(dx), L.ptr(gate), L.ACT_LRELU, C.c_float(0.0), L.ptr(g),
                                 C.c_int64(dx.numel()), L.stream_ptr()), "evt_dact_mul")
        dx = g if gate_pos == 1.0 else g.mul_(gate_pos)
    if add is not None:
        dx.add_(add)
    return dx


def gemm_bwd_weight(slot, x, dy, want_bias=True):
    """dW += dy^T x (and db += column sums of dy) straight into the parameters' fp32 arena views when the engine
    attached them (`_evt_grad_view`): returns (None, None) then; otherwise (dw [N, K], db [N] or None) fp32 tensors."""
    M = x.numel() // slot.K
    p = slot.params(M, False)
    wv = getattr(slot.weight, "_evt_grad_view", None)
    want_b = slot.bias is not None and want_bias
    bv = getattr(slot.bias, "_evt_grad_view", None) if want_b else None
    sunk = (wv is not None and slot.Np == slot.N and wv.dtype == torch.float32 and wv.is_contiguous()
            and tuple(wv.shape) == (slot.N, slot.K) and (not want_b or (bv is not None and bv.numel() == slot.N)))
    if sunk:
        dw_buf, db_buf = wv, bv
    else:
        dw_buf = torch.zeros((slot.Np, slot.K), dtype=torch.float32, device=dy.device)
        db_buf = torch.zeros(slot.Np, dtype=torch.float32, device=dy.device) if want_b else None
    bank = slot.bank
    if sunk and bank.side_on and bank.side is not None and x.is_cuda and not torch.cuda.is_current_stream_capturing():
        # straight into the arena, on the side stream: ordered behind what the current stream has enqueued (x, dy exist),
        # the operands kept alive for the allocator until the launch has run; S1Engine joins before it reads the arena
        if bank.pair:
            bank._pending.append((p, x, dy, dw_buf, db_buf))
            bank._side_used = True
            if len(bank._pending) >= 8:          # no attention backward in sight (a stack of plain linears)
                bank.flush_pending()
            return None, None
        bank.side.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(bank.side):
            L.check(L.lib().evt_gemm_bf16_bwd_weight(C.byref(p), L.ptr(x), L.ptr(dy), L.ptr(dw_buf), L.ptr(db_buf),
                                                     L.stream_ptr()), "evt_gemm_bf16_bwd_weight")
        x.record_stream(bank.side)
        dy.record_stream(bank.side)
        bank._side_used = True
        return None, None
    L.check(L.lib().evt_gemm_bf16_bwd_weight(C.byref(p), L.ptr(x), L.ptr(dy), L.ptr(dw_buf), L.ptr(db_buf),
                                             L.stream_ptr()), "evt_gemm_bf16_bwd_weight")
    if sunk:
        return None, None
    dw, db = dw_buf, db_buf
    if slot.Np != slot.N:
        dw = dw[:slot.N]
        db = db[:slot.N] if db is not None else None
    return dw, db


class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b) on the last dimension; x [..., K] contiguous in the bank's dtype -> y [..., Np]."""

    @staticmethod
    def forward(ctx, x, weight, bias, anchor, slot, relu):
        y = gemm_fwd(slot, x, relu=relu)
        ctx.slot, ctx.relu = slot, relu
        ctx.save_for_backward(x, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        slot = ctx.slot
        dy = dy.contiguous()
        if ctx.relu:
            dy_eff = torch.empty_like(dy)
            L.check(L.lib().evt_dact_mul(L.dt_of(dy), L.ptr(dy), L.ptr(y), L.ACT_LRELU, C.c_float(0.0), L.ptr(dy_eff),
                                         C.c_int64(dy.numel()), L.stream_ptr()), "evt_dact_mul")
            dy = dy_eff
        dw = db = None
        if ctx.needs_input_grad[1]:
            # inside an engine the parameters carry the fp32 view of their slot in the flat gradient arena
            # (`_evt_grad_view`): the launch accumulates (+=) straight into it -- across the accumulation micro-batches too --
            # and autograd gets no tensor: no zero-filled [N, K] scratch per layer and micro-step, nothing to add afterwards
            dw, db = gemm_bwd_weight(slot, x, dy, want_bias=ctx.needs_input_grad[2])
        dx = gemm_bwd_data(slot, dy) if ctx.needs_input_grad[0] else None
        return dx, dw, db, None, None, None


def linear(x, weight, bias=None, relu=False):
    """F.linear(x, weight, bias) (+ relu) through the bank the weight is attached to.  Returns [..., Np]: the output
    width padded to a multiple of 128 when N is not a multiple of 8 (only the 1025-entry vocabulary projection), columns
    >= N being zero.  There is no fallback: a weight without a bank raises."""
    slot = getattr(weight, "_evt_slot", None)
    if slot is None:
        raise L.EvtError("linear(): the weight is not attached to a LinearBank (S1Engine builds it); no eager fallback")
    return LinearFn.apply(x, weight, bias, slot.bank.anchor, slot, bool(relu))
