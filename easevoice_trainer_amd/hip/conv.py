"""Host side of the fused Conv1d / ConvTranspose1d family: parameter modules with the reference's
state_dict keys, the prepared-weight bank, and the autograd Functions that call the C ABI.

Layout contract: activations are channels-last [nseq, L, C] (contiguous); the reference's [B, C, L]
tensors are transposed once at the model boundary.  Reference modules mirrored:
torch.nn.Conv1d / ConvTranspose1d / Conv2d((k,1)) with optional old-style weight_norm
(`weight_g` / `weight_v` keys), src/easevoice/module/models.py:419-446,486-536,563-574,
modules.py:154-185,226-296.
"""
import ctypes as C
import os
import math

import torch
from torch import nn

from . import lib as L


class ConvSlot:
    """Per-conv record inside a WeightBank: geometry + views into the prepared-weight arenas."""

    __slots__ = ("module", "layout", "reg", "alt", "dw", "bank", "_pcache", "packed_member", "parts", "dw_extra",
                 "db_part", "used", "wg_used", "wg_dirty", "_wgp")

    def __init__(self, module, layout, bank):
        self.module, self.layout, self.bank = module, layout, bank
        self.reg = self.alt = self.dw = None
        self._pcache = {}
        self.packed_member = False   # True: the weight gradient of this conv is produced by a PackedConv over it
        # deterministic split-K (evt_conv1d_bwd_weight_parts): slabs of the gradient image beyond `dw`, partial bias
        # gradients, the device counters the kernels write, and the host's copy of the slab count of this step
        self.parts, self.dw_extra, self.db_part, self.used, self.wg_used, self._wgp = 1, None, None, None, 0, None
        self.wg_dirty = False        # a weight-gradient launch of this step has written into `dw` already

    def params(self, nseq, lin, in_slope, out_act, out_slope):
        key = (nseq, lin, in_slope, out_act, out_slope, self.bank.impl)
        p = self._pcache.get(key)
        if p is None:
            m = self.module
            p = L.ConvParams(self.bank.dt, nseq, lin, m.cin, m.cout, m.k, m.stride, m.pad, m.dil, m.groups,
                             1 if m.transposed else 0, in_slope, out_act, out_slope, self.bank.impl)
            self._pcache[key] = p
        return p


def conv_layout(dt, cin, cout, k, stride, pad, dil, groups, transposed) -> L.WLayout:
    p = L.ConvParams(dt, 1, max(64, k * dil + 1), cin, cout, k, stride, pad, dil, groups, 1 if transposed else 0,
                     1.0, 0, 1.0, 0)
    lay = L.WLayout()
    L.check(L.lib().evt_conv1d_layout(C.byref(p), C.byref(lay)), "evt_conv1d_layout")
    return lay


class EvtConv1d(nn.Module):
    """Conv1d / ConvTranspose1d parameter holder.  Keys: `weight` (+`bias`) or `weight_g`/`weight_v`
    (+`bias`) exactly as torch's (old-style weight-normed) modules expose them; `kdims=2` appends the
    trailing size-1 kernel dim of DiscriminatorP's Conv2d((k,1)) weights; `kdims=0` (k = 1 only) drops the kernel dim:
    the `weight` [cout, cin] / `bias` of an nn.Linear, run as a 1x1 convolution over the rows."""

    def __init__(self, cin, cout, k, stride=1, padding=0, dilation=1, groups=1, bias=True, transposed=False,
                 weight_norm=False, kdims=1):
        super().__init__()
        self.cin, self.cout, self.k, self.stride, self.pad, self.dil = cin, cout, k, stride, padding, dilation
        self.groups, self.transposed, self.weight_norm, self.kdims = groups, transposed, weight_norm, kdims
        d0 = cin if transposed else cout
        d1 = cout if transposed else cin // groups
        if kdims == 0 and (k != 1 or transposed or groups != 1):
            raise L.EvtError("kdims=0 is the nn.Linear layout: k = 1, dense, not transposed")
        wshape = (d0, d1) if kdims == 0 else (d0, d1, k) + ((1,) if kdims == 2 else ())
        w = torch.empty(wshape)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        fan_in = d1 * k if not transposed else d0 * k  # torch: fan_in is computed from weight.size(1)*k
        fan_in = w.size(1) * k
        bias_p = None
        if bias:
            bound = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0
            bias_p = nn.Parameter(torch.empty(cout).uniform_(-bound, bound))
        # registration order = the reference's parameters() order, which numbers the optimiser state of a checkpoint:
        # plain conv (weight, bias); torch.nn.utils.weight_norm removes `weight` and appends weight_g, weight_v, so a
        # weight-normed conv enumerates as (bias, weight_g, weight_v)
        if weight_norm:
            self.register_parameter("bias", bias_p)
            self.weight_g = nn.Parameter(w.reshape(d0, -1).norm(dim=1).reshape((d0,) + (1,) * (w.dim() - 1)))
            self.weight_v = nn.Parameter(w)
        else:
            self.weight = nn.Parameter(w)
            self.register_parameter("bias", bias_p)
        self._slot = None

    @property
    def v(self):
        return self.weight_v if self.weight_norm else self.weight

    def lout(self, lin):
        if not self.transposed:
            return (lin + 2 * self.pad - self.dil * (self.k - 1) - 1) // self.stride + 1
        return (lin - 1) * self.stride - 2 * self.pad + self.dil * (self.k - 1) + 1

    def forward(self, x, res=None, in_slope=1.0, out_act=L.ACT_NONE, out_slope=1.0):
        if self._slot is None:
            raise L.EvtError("EvtConv1d used before WeightBank.attach(); there is no eager fallback")
        return ConvFn.apply(x, self._slot.bank.anchor, res, self._slot, float(in_slope), int(out_act),
                            float(out_slope))


class PackedConv:
    """Several equal-shaped k = 1 convolutions of ONE input as a single convolution with the concatenated output channels
    (the q | k | v projections of an attention layer: three [C, C] weights -> one dense [3C, C] GEMM, one launch each way
    instead of three).  Exists only when the members' parameters are adjacent in the runtime's arena (ParamArena honours
    the owner's `arena_adjacent()`): `v` / `bias` and their gradients are then plain views over the members' storage,
    so the packed dW image unfolds straight into the members' .grad -- nothing is copied, the members stay usable."""
    weight_norm, kdims, transposed, groups, stride, pad, dil, k = False, 1, False, 1, 1, 0, 1, 1

    def __init__(self, members):
        m0 = members[0]
        self.members = members
        self.cin, self.cout = m0.cin, sum(m.cout for m in members)
        n, es = m0.weight.numel(), m0.weight.element_size()
        ws, bs = [m.weight for m in members], [m.bias for m in members]

        def adjacent(ts, numel):
            return all(t.data_ptr() == ts[0].data_ptr() + i * numel * es and t.is_contiguous() for i, t in enumerate(ts))

        ok = (all(m.cin == m0.cin and m.cout == m0.cout and m.k == 1 and not m.weight_norm and m.kdims == 1 and
                  m.bias is not None for m in members) and adjacent(ws, n) and adjacent(bs, m0.cout)
              and all(t.grad is not None for t in ws + bs) and adjacent([t.grad for t in ws], n)
              and adjacent([t.grad for t in bs], m0.cout))
        self.ok = ok
        if ok:
            self.v = torch.as_strided(ws[0].data, (self.cout, self.cin, 1), (self.cin, 1, 1))
            self.v.grad = torch.as_strided(ws[0].grad, (self.cout, self.cin, 1), (self.cin, 1, 1))
            self.bias = torch.as_strided(bs[0].data, (self.cout,), (1,))
            self.bias.grad = torch.as_strided(bs[0].grad, (self.cout,), (1,))
        self._slot = None

    def lout(self, lin):
        return lin


def slab_count(image_bytes, d0, budget=48 << 20, cap=32):
    """slabs of one convolution's gradient image for the deterministic split-K weight gradient (= the most position splits
    its kernel may use).  A small image can afford many (the reduction of a 128 -> 128 k11 vocoder layer runs over 40960
    positions for 180 K outputs); a 21 MB image (1024 -> 1024 k5) gets 3.  `cap` (32; measured on the s2 step: 26.2 ms,
    64: 26.5, 8: 30.9, 2: 52) bounds what evt_wn_grad_multi has to fold per image; two kinds of layer go beyond it, because
    all their parallelism comes from the split: images of a few KB (the 16- / 32-channel vocoder stages: 11 - 45 KB for
    160 K - 330 K positions) and layers with at most 64 output channels (one or two output tiles: 64 -> 64 k11 over 81920
    positions, in the step: 66 us with 32 slabs, 23 us with 128).  The s2 generator's slabs come to 5.2 GB, the
    discriminators' to 0.7 GB, of the 288 GB."""
    limit = 256 if image_bytes <= (64 << 10) else (128 if d0 <= 64 else cap)
    return max(1, min(limit, -(-budget // max(image_bytes, 1))))


class WeightBank:
    """All prepared conv weights of one model: REG/ALT images in the compute dtype, fp32 dW images,
    and the device tables for the two multi-tensor launches (fold before forward, grad after backward)."""

    def __init__(self, model: nn.Module, dtype: torch.dtype, device, impl=L.IMPL_AUTO):
        L.set_half(dtype)     # a 16-bit compute dtype selects the build of the library that serves it
        self.dt = L.dt_code(dtype)
        self.dtype, self.device, self.impl = dtype, torch.device(device), impl
        self.weight_grads = True
        # EVT_ASYNC_WGRAD=1: weight gradients on a side HIP stream next to the backward-data chain; grads() joins and
        # the operands are held until then.  Measured on MI355X: -1.6 ms/step with eager launches, +7 ms/step under
        # HIP-graph replay (fork/join edges per conv), so it is off by default.
        self.async_wgrad = self.device.type == "cuda" and os.environ.get("EVT_ASYNC_WGRAD", "0") == "1"
        # EVT_WGRAD_DEFER=N (default 64; 0 = off): weight-gradient launches are queued and handed to a side HIP stream N at
        # a time -- ONE fork per batch instead of one per convolution -- so the backward-data chain (the critical path:
        # every launch waits for the one before it) shares the chip with the weight gradients, which depend on nothing
        # but their two operands.  grads() joins.  The operands are held until the join.  Measured on MI355X, s2 step
        # under HIP-graph replay, three interleaved rounds: 27.85-27.95 ms without, 27.14-27.18 ms with N = 64; small N
        # (a fork every few convolutions) is what made EVT_ASYNC_WGRAD slower under replay.
        # Round 6, under the branch streams (profiles/r06_streams.txt): the generator's ~150 launches flushed every 48 instead
        # of every 64 is 0.3 ms ahead (40-56 level except 52; 32 and 80-96 behind -- it matters which stretch of the backward a
        # flush lands beside); the discriminators' 42 stay one flush at the end either way (in pieces: +0.2 ms).
        self.defer_n = int(os.environ.get("EVT_WGRAD_DEFER", "48")) if self.device.type == "cuda" else 0
        self._deferred = []
        self._deferred_on = set()
        self._side = None
        self._held = []
        self.anchor = torch.zeros(1, device=device, requires_grad=True)
        self.slots = []
        reg_n = alt_n = 0
        ALIGN = 128  # elements; keeps every image 256-byte aligned
        offs = []
        convs = [m for m in model.modules() if isinstance(m, EvtConv1d)]
        # packed projections (see PackedConv): only for the bf16 bank -- the fused attention node that uses them is bf16
        for m in model.modules():
            fn = getattr(m, "qkv_pack_modules", None)
            members = fn() if (fn is not None and L.is_half(dtype)) else None
            m_packed = None
            if members:
                pc = PackedConv(members)
                if pc.ok:
                    convs.append(pc)
                    m_packed = pc
            if fn is not None:
                m._qkv_packed = m_packed
        for m in convs:
            lay = conv_layout(self.dt, m.cin, m.cout, m.k, m.stride, m.pad, m.dil, m.groups, m.transposed)
            s = ConvSlot(m, lay, self)
            m._slot = s
            self.slots.append(s)
            offs.append((reg_n, alt_n))
            reg_n += (lay.reg_elems + ALIGN - 1) // ALIGN * ALIGN
            alt_n += (lay.alt_elems + ALIGN - 1) // ALIGN * ALIGN
            if isinstance(m, PackedConv):
                for mm in m.members:
                    mm._slot.packed_member = True
        self.reg_arena = torch.zeros(max(reg_n, 1), dtype=dtype, device=device)
        self.alt_arena = torch.zeros(max(alt_n, 1), dtype=dtype, device=device)
        # slab 0 of every gradient image + the kernels' slab counters (two int32 per conv): ONE memset per step clears both
        ns = len(self.slots)
        self.dw_arena = torch.zeros(max(reg_n, 1) + 2 * ns, dtype=torch.float32, device=device)
        self.used_all = self.dw_arena[max(reg_n, 1):].view(torch.int32)
        # slabs 1.. (EVT_WGRAD_PARTS=0 switches the deterministic split off: fp32 atomics into the one image as before);
        # how many per image: slab_count().  Never zeroed: a slab is stored before it is read, the counters say how many
        # are valid.
        self.parts_on = (L.is_half(dtype) and self.device.type == "cuda"
                         and os.environ.get("EVT_WGRAD_PARTS", "1") != "0")
        # the slab sizes below (48 MiB per image, at most 32 slabs: 5.9 GB for the s2 models) were measured on a 288 GB
        # MI355X; on a smaller device the budget shrinks with its memory (at 1/4 of the memory: 12 MiB, 1.5 GB)
        total_mem = torch.cuda.get_device_properties(self.device).total_memory if self.device.type == "cuda" else 0
        mem_scale = min(1.0, total_mem / float(256 << 30)) if total_mem else 1.0
        budget = int(os.environ.get("EVT_WGRAD_SLAB_MB", str(max(4, int(48 * mem_scale))))) << 20
        # slabs per image: more slabs = more split-K blocks for the weight-gradient launches, and as many more bytes for the
        # fold at the end of the backward (wn_grad).  32 until the launches moved off the critical path (side stream, branch
        # streams); since then the fold's bytes weigh more: 16-24 measured level and 0.2 ms ahead of 32, 8 and 4 behind
        # (profiles/r06_streams.txt)
        cap = int(os.environ.get("EVT_WGRAD_PARTS_CAP", "20"))
        # operands held for the deferred weight-gradient launches: flushed on a byte budget as well as on a count
        self.defer_bytes = int(os.environ.get("EVT_WGRAD_DEFER_MB", str(max(256, int(4096 * mem_scale))))) << 20
        self._deferred_bytes = 0
        ex_n = db_n = 0
        ex_offs = []
        for s in self.slots:
            s.parts = slab_count(s.layout.reg_elems * 4, s.layout.d0, budget, cap) if self.parts_on else 1
            ex_offs.append((ex_n, db_n))
            ex_n += (s.parts - 1) * ((s.layout.reg_elems + ALIGN - 1) // ALIGN * ALIGN)
            db_n += s.parts * ((s.layout.d0 + 31) // 32 * 32)
        # scratch of the two-launch reductions (evt_wgrad_parts.ws): one buffer per stream the weight gradients run on
        self._ws = {}
        # zeros once: the kernels never store the padded taps of an image row, and the fold adds whole rows
        self.dw_extra_arena = torch.zeros(max(ex_n, 1), dtype=torch.float32, device=device) if self.parts_on else None
        self.db_part_arena = torch.zeros(max(db_n, 1), dtype=torch.float32, device=device) if self.parts_on else None
        for i, (s, (ro, ao)) in enumerate(zip(self.slots, offs)):
            s.reg = self.reg_arena[ro: ro + s.layout.reg_elems]
            s.alt = self.alt_arena[ao: ao + s.layout.alt_elems]
            s.dw = self.dw_arena[ro: ro + s.layout.reg_elems]
            s.used = self.used_all[2 * i: 2 * i + 2]
            if self.parts_on:
                eo, do = ex_offs[i]
                stride = (s.layout.reg_elems + ALIGN - 1) // ALIGN * ALIGN
                s.dw_extra = self.dw_extra_arena[eo: eo + (s.parts - 1) * stride]
                s.db_part = self.db_part_arena[do: do + s.parts * s.layout.d0]
        self._items = self._rows = None
        self._nrows = 0
        # images that also exist in MFMA fragment order (frag()): re-made behind every fold by one launch
        self._frags, self._frag_items, self._frag_rows, self._frag_tables = {}, [], [], {}
        self._frag_retired = []       # superseded device tables, kept alive for graphs captured with them

    def build_tables(self):
        """(Re)build the device descriptor tables.  Must be called after parameters (and their .grad
        buffers) have reached their final storage (ParamArena.flatten)."""
        items, rows, groups = [], [], []
        self._slot_rows = []
        self._group_of_row = {}       # first table row of a slot -> first group of that slot (and the end sentinel)
        for i, s in enumerate(self.slots):
            m = s.module
            self._slot_rows.append((len(rows), len(rows) + s.layout.d0))
            v = m.v
            if v.grad is None:
                v.grad = torch.zeros_like(v)
            g = m.weight_g if m.weight_norm else None
            if g is not None and g.grad is None:
                g.grad = torch.zeros_like(g)
            it = L.WPrepItem()
            it.v = v.data_ptr()
            it.g = g.data_ptr() if g is not None else None
            # a packed member's .grad memory is updated by the pack's rows: two blocks doing `dv += ...` on the same
            # addresses would race (lost update), so the member's own rows are switched off in evt_wn_grad_multi
            it.reg, it.alt, it.dw = s.reg.data_ptr(), s.alt.data_ptr(), (None if s.packed_member else s.dw.data_ptr())
            it.dv = v.grad.data_ptr()
            it.dg = g.grad.data_ptr() if g is not None else None
            it.lay = s.layout
            it.dtype = self.dt
            it.src_d1 = int(getattr(m, "src_d1", 0))
            if self.parts_on and not s.packed_member:
                it.dw_extra = s.dw_extra.data_ptr() if s.parts > 1 else None
                it.dw_part_stride = (s.layout.reg_elems + 127) // 128 * 128
                it.used = s.used.data_ptr()
                bias = getattr(m, "bias", None)
                if bias is not None and not m.transposed:      # fused bias gradients exist for plain convolutions only
                    if bias.grad is None:
                        bias.grad = torch.zeros_like(bias)
                    it.db_part, it.db = s.db_part.data_ptr(), bias.grad.data_ptr()
            items.append(it)
            self._group_of_row[len(rows)] = len(groups)
            rows.extend((i, r) for r in range(s.layout.d0))
            # the fold walks groups of eight consecutive rows (evt_wn_fold_groups): both images leave as 16-byte pieces
            groups.extend((i, r, min(8, s.layout.d0 - r)) for r in range(0, s.layout.d0, 8))
        self._group_of_row[len(rows)] = len(groups)
        self._items = L.struct_to_device(items, self.device)
        self._rows = torch.tensor(rows, dtype=torch.int32, device=self.device).contiguous()
        self._groups = torch.tensor(groups, dtype=torch.int32, device=self.device).contiguous()
        self._nrows = len(rows)
        self._grad_stamp = self._grad_ptrs()

    def _grad_ptrs(self):
        """addresses of every .grad the device tables point at (weights, gains, fused bias gradients)"""
        out = []
        for s in self.slots:
            m = s.module
            g = m.weight_g if m.weight_norm else None
            b = getattr(m, "bias", None)
            out.append((m.v.grad.data_ptr() if m.v.grad is not None else 0,
                        g.grad.data_ptr() if (g is not None and g.grad is not None) else 0,
                        b.grad.data_ptr() if (b is not None and b.grad is not None) else 0))
        return out

    def fold(self, lo=None, hi=None):
        """w = g*v/|v| -> REG/ALT images for every conv of the model: ONE launch (or the rows [lo, hi) of rows_of())."""
        if self._items is None:
            self.build_tables()
        lo = 0 if lo is None else lo
        hi = self._nrows if hi is None else hi
        if hi <= lo:
            return
        glo, ghi = self._group_of_row.get(lo), self._group_of_row.get(hi)
        if glo is not None and ghi is not None and os.environ.get("EVT_FOLD_GROUPS", "1") != "0":
            # row ranges start and end at layer boundaries (rows_of): the same rows as groups of eight
            grp = C.c_void_p(self._groups.data_ptr() + 12 * glo)    # (item, first row, rows) int32 triples
            L.check(L.lib().evt_wn_fold_groups(L.ptr(self._items), grp, ghi - glo, L.stream_ptr()), "evt_wn_fold_groups")
        else:
            rows = C.c_void_p(self._rows.data_ptr() + 8 * lo)       # (item, row) int32 pairs
            L.check(L.lib().evt_wn_fold_multi(L.ptr(self._items), rows, hi - lo, L.stream_ptr()), "evt_wn_fold_multi")
        self._repack_frags(lo, hi)

    def frag(self, slot, which):
        """`slot`'s REG (which = "reg") or ALT ("alt") image in MFMA FRAGMENT ORDER (evt_frag_pack: [16-row tile][K step of
        32][64 lanes][8], lane (n, g) = row 16 tile + n, K elements 32 ks + 8 g ..): what the kernels that stream a weight
        image straight from global memory into MFMA operands with ONE MFMA per fragment read (csrc/wn_layer.hip) -- a
        wave's fragment load is then 1 KiB contiguous instead of 16 rows x 64 bytes, which the vector memory path serves at
        ~12 B/clk per CU (34 -> 20 us per WN layer).  Tried for csrc/resunit_wide.hip too (9-14 MFMAs per fragment): no change
        (23.06-23.20 against 22.96-22.98 ms per step), its loads are not what it waits for -- it reads REG / ALT directly.  A copy owned by the bank: made on first request, re-made behind every fold of the slot's rows (ONE launch for
        all registered images, captured with the fold when the step is a HIP graph)."""
        key = (id(slot), which)
        f = self._frags.get(key)
        if f is None:
            lay = slot.layout
            if which == "reg":
                rows, ktot, src, elems = lay.d0, lay.reg_nchunk * lay.reg_kp * lay.reg_ck, slot.reg, lay.reg_elems
            else:
                rows, ktot, src, elems = lay.d1, lay.alt_nchunk * lay.alt_kp * lay.alt_ck, slot.alt, lay.alt_elems
            if elems != rows * ktot or rows % 16 or ktot % 32 or not L.is_half(self.dtype):
                raise L.EvtError(f"no fragment-order form for this image ({which}, {rows} x {ktot}, {self.dtype})")
            if self._items is None:
                self.build_tables()
            f = torch.empty(elems, dtype=self.dtype, device=self.device)
            it = L.FragItem()
            it.src, it.dst, it.rows, it.ktot = src.data_ptr(), f.data_ptr(), rows, ktot
            self._frags[key] = f
            self._frag_items.append(it)
            self._frag_rows.append(self._slot_rows[self.slots.index(slot)])
            # a new image changes what a fold of its rows has to re-pack: the row-range tables are rebuilt on demand, but
            # the old device tables stay ALIVE -- a HIP graph captured earlier holds their addresses
            self._frag_retired.extend(t for t, _n in self._frag_tables.values() if t is not None)
            self._frag_tables.clear()
            one = L.struct_to_device([it], self.device)
            self._frag_tables[("one", len(self._frag_items))] = (one, 1)      # kept alive
            L.check(L.lib().evt_frag_pack(L.ptr(one), 1, L.stream_ptr()), "evt_frag_pack")
            # the whole-model table is rebuilt NOW (images are registered by the first eager forward), so that the fold a
            # later graph capture records finds it instead of building it -- a host-to-device copy -- inside the capture
            if not torch.cuda.is_current_stream_capturing():
                items = list(self._frag_items)
                self._frag_tables[(0, self._nrows)] = (L.struct_to_device(items, self.device), len(items))
        return f

    def _repack_frags(self, lo, hi):
        if not self._frag_items:
            return
        tab = self._frag_tables.get((lo, hi))
        if tab is None:
            items = [it for it, (r0, r1) in zip(self._frag_items, self._frag_rows) if r0 < hi and r1 > lo]
            tab = self._frag_tables[(lo, hi)] = (L.struct_to_device(items, self.device) if items else None, len(items))
        if tab[1]:
            L.check(L.lib().evt_frag_pack(L.ptr(tab[0]), tab[1], L.stream_ptr()), "evt_frag_pack")

    def zero_dw(self):
        if self._deferred or self._held:
            # weight-gradient launches of a backward that never reached grads() (a standalone user, an exception in the
            # middle of a step): they would land in the images zeroed below -- drop them, wait for what already runs
            self._deferred.clear()
            self._deferred_on.clear()
            self._deferred_bytes = 0
            if self._side is not None:
                torch.cuda.current_stream(self.device).wait_stream(self._side)
            self._held.clear()
        self.dw_arena.zero_()            # slab 0 of every image and the slab counters
        self._tables_checked = False     # the gradient tensors are looked at once per step, by the first grads() call
        for s in self.slots:
            s.wg_used, s.wg_dirty = 0, False

    def scratch(self):
        """16 MiB of scratch for the current stream (created on first use)"""
        key = torch.cuda.current_stream(self.device).cuda_stream
        ws = self._ws.get(key)
        if ws is None:
            ws = self._ws[key] = torch.empty(4 << 20, dtype=torch.float32, device=self.device)
        return ws

    def side_stream(self):
        if self._side is None:
            self._side = L.role_stream(self.device, "bank")
        return self._side

    def flush_deferred(self):
        """hand the queued weight-gradient launches to the side stream (one fork)"""
        if not self._deferred:
            return
        side = self.side_stream()
        side.wait_stream(torch.cuda.current_stream(self.device))
        for st in self._deferred_on:          # operands queued from another stream than the one that flushes
            side.wait_stream(st)
        self._deferred_on.clear()
        with torch.cuda.stream(side):
            for args in self._deferred:
                _bwd_weight_now(*args)
        self._held.extend((a[1], a[2], a[3]) for a in self._deferred)
        self._deferred.clear()
        self._deferred_bytes = 0

    def join_side(self):
        self.flush_deferred()
        if self._side is not None and self._held:
            torch.cuda.current_stream(self.device).wait_stream(self._side)
        self._held.clear()

    def rows_of(self, modules):
        """[lo, hi) of the row table covered by `modules` (EvtConv1d instances of this bank, contiguous in slot order):
        lets a data-parallel step finish -- and start reducing -- the gradients of one sub-model while the backward of
        the next is still running"""
        idx = sorted(i for i, s in enumerate(self.slots) if any(s.module is m for m in modules))
        if not idx or idx != list(range(idx[0], idx[-1] + 1)):
            raise L.EvtError("rows_of: the modules must be a contiguous, non-empty run of this bank's convolutions")
        if self._items is None:
            self.build_tables()
        return self._slot_rows[idx[0]][0], self._slot_rows[idx[-1]][1]

    def check_tables(self):
        """a .grad replaced since the tables were built (zero_grad(set_to_none=True), a caller assigning a new tensor) would
        leave the fused bias gradients / dv / dg writing to the old storage -- rebuild the tables.  Looked at ONCE per step
        (zero_dw() re-arms it): the data-parallel program calls grads() per sub-model, and walking every slot's three
        pointers each time was pure host time on a launch-bound path.  A capture validates before it records: the check
        runs on the eager steps that precede it, and a replaced tensor between capture and replay would invalidate the
        graph's other recorded pointers as well."""
        if self._items is None:
            self.build_tables()
        elif not getattr(self, "_tables_checked", False) and not torch.cuda.is_current_stream_capturing():
            if self._grad_stamp != self._grad_ptrs():
                self.build_tables()
            self._tables_checked = True

    def grads(self, lo=None, hi=None, join=True):
        """dW images -> weight_v.grad / weight_g.grad (or weight.grad), ACCUMULATED (+=): one launch over the whole
        model, or over the row range [lo, hi) of rows_of() -- every row must be visited exactly once per backward.
        join=False: the caller's stream already waits for the weight-gradient side stream (ModelRuntime.book_piece)."""
        if join:
            self.join_side()
        self.check_tables()
        lo = 0 if lo is None else lo
        hi = self._nrows if hi is None else hi
        if hi <= lo:
            return
        rows = C.c_void_p(self._rows.data_ptr() + 8 * lo)       # (item, row) int32 pairs
        L.check(L.lib().evt_wn_grad_multi(L.ptr(self._items), rows, hi - lo, L.stream_ptr()), "evt_wn_grad_multi")


class FrozenConv:
    """geometry + tensor getters of a convolution whose weight is never trained (ssl_proj, the RVQ codebook as a 1x1
    layer): what ConvSlot / _fwd need from a module, nothing that WeightBank would pick up from model.modules()"""
    weight_norm, kdims, transposed, groups, dil = False, 1, False, 1, 1

    def __init__(self, weight_fn, bias_fn, cin, cout, k=1, stride=1, padding=0):
        self.weight_fn, self.bias_fn = weight_fn, bias_fn
        self.cin, self.cout, self.k, self.stride, self.pad = cin, cout, k, stride, padding
        self._slot = None

    @property
    def v(self):
        w = self.weight_fn()
        return w if w.dim() == 3 else w.unsqueeze(-1)

    @property
    def bias(self):
        return self.bias_fn() if self.bias_fn is not None else None

    def lout(self, lin):
        return (lin + 2 * self.pad - (self.k - 1) - 1) // self.stride + 1


class FrozenBank:
    """fp32 images of weights outside the trained bank -- the s2 quantizer path runs in fp32 whatever the compute dtype of
    the step (models.py:912-921 disables autocast there).  Folded on first use and again whenever the tensors were
    replaced or written through torch (load_state_dict, the k-means initialisation): torch's version counters tell."""
    weight_grads = False

    def __init__(self, convs, device, impl=L.IMPL_AUTO):
        self.dt, self.dtype, self.device, self.impl = L.DT_F32, torch.float32, torch.device(device), impl
        self.slots, offs, reg_n, alt_n = [], [], 0, 0
        for m in convs:
            lay = conv_layout(self.dt, m.cin, m.cout, m.k, m.stride, m.pad, m.dil, m.groups, m.transposed)
            s = ConvSlot(m, lay, self)
            m._slot = s
            self.slots.append(s)
            offs.append((reg_n, alt_n))
            reg_n += (lay.reg_elems + 127) // 128 * 128
            alt_n += (lay.alt_elems + 127) // 128 * 128
        self.reg_arena = torch.zeros(max(reg_n, 1), dtype=torch.float32, device=device)
        self.alt_arena = torch.zeros(max(alt_n, 1), dtype=torch.float32, device=device)
        for s, (ro, ao) in zip(self.slots, offs):
            s.reg = self.reg_arena[ro: ro + s.layout.reg_elems]
            s.alt = self.alt_arena[ao: ao + s.layout.alt_elems]
        self._stamp = self._items = None

    def prepare(self):
        vs = [s.module.v for s in self.slots]
        stamp = tuple((v.data_ptr(), v._version) for v in vs)
        if stamp == self._stamp:
            return
        items, rows = [], []
        for i, (s, v) in enumerate(zip(self.slots, vs)):
            if v.dtype != torch.float32 or not v.is_contiguous() or v.device != self.device:
                raise L.EvtError("FrozenBank: contiguous fp32 weights on the bank's device expected")
            it = L.WPrepItem()
            it.v, it.g, it.reg, it.alt = v.data_ptr(), None, s.reg.data_ptr(), s.alt.data_ptr()
            it.dw = it.dv = it.dg = None
            it.lay, it.dtype = s.layout, self.dt
            items.append(it)
            rows.extend((i, r) for r in range(s.layout.d0))
        self._items = L.struct_to_device(items, self.device)
        self._rows = torch.tensor(rows, dtype=torch.int32, device=self.device).contiguous()
        L.check(L.lib().evt_wn_fold_multi(L.ptr(self._items), L.ptr(self._rows), len(rows), L.stream_ptr()),
                "evt_wn_fold_multi")
        self._stamp = stamp


PLAIN_X = os.environ.get("EVT_CONV_PLAIN_X", "1") != "0"   # A/B switch: pre-activated input for GEMM-sized layers with a load-side leaky-relu
TRACE = None   # profiling only (set_trace): (tag, kind, flops, bytes, ev0, ev1, shape, module) per launch


TRACE_RANGES = False   # each traced launch also opens a torch.profiler range "evt#<record index>"


def set_trace(rec, ranges=False):
    """profiling aid of bench.py's roofline leg: a list switches per-launch records (and the library's kernel-name tags)
    on, None switches both off (the product state).  ranges=True wraps every traced launch in a torch.profiler
    record_function range, so that a surrounding torch.profiler session attributes the launch's kernels (durations from
    the same tracer rocprofv3 uses) to the record -- HIP-event pairs add a few microseconds to every small launch."""
    global TRACE, TRACE_RANGES
    TRACE = rec
    TRACE_RANGES = bool(ranges) and rec is not None
    L.lib().evt_debug_kernel_tags(1 if rec is not None else 0)


def _t0():
    if TRACE is None:
        return None
    rf = None
    if TRACE_RANGES:
        rf = torch.profiler.record_function(f"evt#{len(TRACE)}")
        rf.__enter__()
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e, rf


def _t1(e0, kind, m, nseq, lin, extra_elems):
    e0, rf = e0
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    if rf is not None:
        rf.__exit__(None, None, None)
    lq = lin if m.transposed else m.lout(lin)
    macs = nseq * lq * m.cin * m.cout * m.k // m.groups
    sz = 2 if L.is_half(m._slot.bank.dtype) else 4
    act = nseq * (lin * m.cin + m.lout(lin) * m.cout) + extra_elems
    wbytes = m.v.numel() * (4 if kind == "bwd_weight" else sz)
    shape = (f"{'T' if m.transposed else ''}{m.cin}>{m.cout} k{m.k} s{m.stride} d{m.dil} g{m.groups} "
             f"n{nseq} L{lin}")
    TRACE.append((L.lib().evt_last_kernel_tag().decode(), kind, 2 * macs, act * sz + wbytes, e0, e1, shape, m))


def _t1_elt(e0, name, nbytes, owner):
    """trace record of an element-wise launch of this file (leaky-relu copy, stage mean, activation derivative): no flops,
    bytes = operands in + out; `owner` = a module of the model the launch belongs to (bench.py attributes HiFi-GAN's
    element-wise launches to `dec` through it)"""
    e0, rf = e0
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    if rf is not None:
        rf.__exit__(None, None, None)
    TRACE.append((name, "elt", 0, nbytes, e0, e1, name, owner))


def _fwd(slot, x, res, in_slope, out_act, out_slope, out=None):
    """`out`: optional preallocated contiguous [nseq, Lout, Cout] destination (e.g. one plane of a q|k|v buffer)"""
    m = slot.module
    if x.dim() != 3 or x.size(2) != m.cin or x.dtype != slot.bank.dtype or not x.is_contiguous():
        raise L.EvtError(f"conv input must be contiguous [nseq, L, {m.cin}] {slot.bank.dtype}, got "
                         f"{tuple(x.shape)} {x.dtype} contiguous={x.is_contiguous()}")
    nseq, lin = x.size(0), x.size(1)
    y = out if out is not None else torch.empty((nseq, m.lout(lin), m.cout), dtype=x.dtype, device=x.device)
    if y.shape != (nseq, m.lout(lin), m.cout) or y.dtype != x.dtype or not y.is_contiguous():
        raise L.EvtError("conv output buffer must be contiguous [nseq, Lout, Cout] in the compute dtype")
    if res is not None and (res.shape != y.shape or not res.is_contiguous() or res.dtype != x.dtype):
        raise L.EvtError("residual must match the output")
    p = slot.params(nseq, lin, in_slope, out_act, out_slope)
    bias = m.bias
    e0 = _t0()
    L.check(L.lib().evt_conv1d_fwd(C.byref(p), L.ptr(x), L.ptr(slot.reg), L.ptr(slot.alt),
                                   L.ptr(bias.data if bias is not None else None), L.ptr(res), L.ptr(y),
                                   L.stream_ptr()), "evt_conv1d_fwd")
    if e0 is not None:
        _t1(e0, "fwd", m, nseq, lin, y.numel() if res is not None else 0)
    return y


def _bwd_data(slot, dy, y, x, dx_add, nseq, lin, in_slope, out_act, out_slope):
    m = slot.module
    dx = torch.empty((nseq, lin, m.cin), dtype=dy.dtype, device=dy.device)
    p = slot.params(nseq, lin, in_slope, out_act, out_slope)
    e0 = _t0()
    L.check(L.lib().evt_conv1d_bwd_data(C.byref(p), L.ptr(dy), L.ptr(y if out_act != L.ACT_NONE else None),
                                        L.ptr(slot.reg), L.ptr(slot.alt), L.ptr(x if in_slope != 1.0 else None),
                                        L.ptr(dx_add), L.ptr(dx), L.stream_ptr()), "evt_conv1d_bwd_data")
    if e0 is not None:
        extra = (dy.numel() if out_act != L.ACT_NONE else 0) + (dx.numel() if in_slope != 1.0 else 0) + \
                (dx.numel() if dx_add is not None else 0)
        _t1(e0, "bwd_data", m, nseq, lin, extra)
    return dx


def _bwd_weight(slot, x, dy, y, nseq, lin, in_slope, out_act, out_slope):
    bank = slot.bank
    if slot.packed_member:
        raise L.EvtError("weight gradient of a packed projection member requested on its own: inside a bf16 runtime the "
                         "q / k / v projections of a windowed attention layer run (and are differentiated) as one pack")
    if bank.defer_n > 0 and TRACE is None:
        bank._deferred.append((slot, x, dy, y, nseq, lin, in_slope, out_act, out_slope))
        bank._deferred_on.add(torch.cuda.current_stream(bank.device))   # a branch stream of the discriminators, or the main one
        bank._deferred_bytes += x.numel() * x.element_size() + dy.numel() * dy.element_size()
        if len(bank._deferred) >= bank.defer_n or bank._deferred_bytes >= bank.defer_bytes:
            bank.flush_deferred()
        return
    if bank.async_wgrad and TRACE is None:
        side = bank.side_stream()
        side.wait_stream(torch.cuda.current_stream(bank.device))
        with torch.cuda.stream(side):
            _bwd_weight_now(slot, x, dy, y, nseq, lin, in_slope, out_act, out_slope)
        bank._held.append((x, dy, y))
        return
    _bwd_weight_now(slot, x, dy, y, nseq, lin, in_slope, out_act, out_slope)


def _bwd_weight_now(slot, x, dy, y, nseq, lin, in_slope, out_act, out_slope):
    m = slot.module
    p = slot.params(nseq, lin, in_slope, out_act, out_slope)
    dbias = None
    if m.bias is not None:
        if m.bias.grad is None:
            m.bias.grad = torch.zeros_like(m.bias)
        dbias = m.bias.grad
    e0 = _t0()
    if slot.bank.parts_on:
        sp = slot._wgp
        if sp is None:
            sp = slot._wgp = L.WgradParts()
            sp.dw_extra = slot.dw_extra.data_ptr() if slot.parts > 1 else None
            sp.part_stride = (slot.layout.reg_elems + 127) // 128 * 128
            sp.db_part = slot.db_part.data_ptr() if (dbias is not None and not m.transposed) else None
            sp.used_dev = slot.used.data_ptr()
            sp.parts = slot.parts
        sp.prev_used, sp.dirty0 = slot.wg_used, int(slot.wg_dirty)
        ws = slot.bank.scratch()
        sp.ws, sp.ws_floats = ws.data_ptr(), ws.numel()
        L.check(L.lib().evt_conv1d_bwd_weight_parts(C.byref(p), L.ptr(x), L.ptr(dy),
                                                    L.ptr(y if out_act != L.ACT_NONE else None), L.ptr(slot.dw),
                                                    L.ptr(dbias), C.byref(sp), L.stream_ptr()),
                "evt_conv1d_bwd_weight_parts")
        slot.wg_used, slot.wg_dirty = max(slot.wg_used, int(sp.used)), True
    else:
        L.check(L.lib().evt_conv1d_bwd_weight(C.byref(p), L.ptr(x), L.ptr(dy),
                                              L.ptr(y if out_act != L.ACT_NONE else None), L.ptr(slot.dw), L.ptr(dbias),
                                              L.stream_ptr()), "evt_conv1d_bwd_weight")
    if e0 is not None:
        _t1(e0, "bwd_weight", m, nseq, lin, dy.numel() if out_act != L.ACT_NONE else 0)


class ConvFn(torch.autograd.Function):
    """y = act_out(conv(lrelu(x, in_slope)) + bias) + res, one HIP launch each way."""

    @staticmethod
    def forward(ctx, x, anchor, res, slot, in_slope, out_act, out_slope):
        ctx.pre = False
        if in_slope != 1.0 and x.is_cuda and x.dim() == 3 and PLAIN_X and L.lib().evt_conv1d_wants_plain_x(
                C.byref(slot.params(x.size(0), x.size(1), in_slope, out_act, out_slope))):
            # GEMM-sized layer with a leaky-relu on load (the vocoder's upsamplers): activate once, the forward and the
            # weight gradient then take plain operands (LDS-DMA kernels); backward-data keeps its fused derivative,
            # read from the sign of the activated tensor
            xa = torch.empty_like(x)
            e0 = _t0()
            L.check(L.lib().evt_leaky_relu(L.dt_of(x), L.ptr(x), C.c_float(in_slope), L.ptr(xa), C.c_int64(x.numel()),
                                           L.stream_ptr()), "evt_leaky_relu")
            if e0 is not None:
                _t1_elt(e0, "lrelu_kernel", 2 * x.numel() * x.element_size(), slot.module)
            y = _fwd(slot, xa, res, 1.0, out_act, out_slope)
            x, ctx.pre = xa, True
        else:
            y = _fwd(slot, x, res, in_slope, out_act, out_slope)
        ctx.slot, ctx.cfg = slot, (in_slope, out_act, out_slope)
        ctx.has_res = res is not None
        ctx.save_for_backward(x, y if out_act != L.ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        slot = ctx.slot
        in_slope, out_act, out_slope = ctx.cfg
        dy = dy.contiguous()
        nseq, lin = x.size(0), x.size(1)
        if out_act != L.ACT_NONE and L.lib().evt_conv1d_wants_plain_dy(
                C.byref(slot.params(nseq, lin, in_slope, out_act, out_slope))):
            # wide layers: apply the activation derivative once, both backward GEMMs then take plain operands
            dy_eff = torch.empty_like(dy)
            e0 = _t0()
            L.check(L.lib().evt_dact_mul(L.dt_of(dy), L.ptr(dy), L.ptr(y), int(out_act), C.c_float(out_slope),
                                         L.ptr(dy_eff), C.c_int64(dy.numel()), L.stream_ptr()), "evt_dact_mul")
            if e0 is not None:
                _t1_elt(e0, "dact_mul_kernel", 3 * dy.numel() * dy.element_size(), slot.module)
            dy, y, out_act, out_slope = dy_eff, None, L.ACT_NONE, 1.0
        if slot.bank.weight_grads:
            # (ctx.pre: x is the activated input -- plain operand for the weight gradient)
            _bwd_weight(slot, x, dy, y, nseq, lin, 1.0 if ctx.pre else in_slope, out_act, out_slope)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _bwd_data(slot, dy, y, x, None, nseq, lin, in_slope, out_act, out_slope)
        dres = dy if (ctx.has_res and ctx.needs_input_grad[2]) else None
        if dres is not None and out_act != L.ACT_NONE:
            raise L.EvtError("residual + output activation: residual is added after the activation, "
                             "its gradient is dy (supported), but was not expected here")
        return dx, None, dres, None, None, None, None


def _resunit_params(s1, s2, x, slope):
    """evt_resunit_params when the fused ResBlock step covers this pair of convolutions and input, else None"""
    m1, m2 = s1.module, s2.module
    if (not L.is_half(x.dtype) or not x.is_cuda or not x.is_contiguous() or x.dim() != 3 or m1.cin != m1.cout
            or m2.cin != m2.cout or m1.cin != m2.cin or m1.k != m2.k or m2.dil != 1 or m1.stride != 1 or m2.stride != 1
            or m1.groups != 1 or m2.groups != 1 or m1.transposed or m2.transposed
            or m1.pad != m1.dil * (m1.k - 1) // 2 or m2.pad != (m2.k - 1) // 2 or s1.bank.impl != L.IMPL_AUTO):
        return None
    p = L.ResUnitParams(L.dt_code(x.dtype), x.size(0), x.size(1), m1.cin, m1.k, m1.dil, float(slope))
    return p if L.lib().evt_resunit_supported(C.byref(p)) else None


def _resunit_wide_params(s1, s2, x, slope):
    """evt_resunit_params when the wide fused step (csrc/resunit_wide.hip: C = 64 / 128) covers this pair, else None"""
    m1, m2 = s1.module, s2.module
    if (not L.is_half(x.dtype) or not x.is_cuda or not x.is_contiguous() or x.dim() != 3 or m1.cin != m1.cout
            or m2.cin != m2.cout or m1.cin != m2.cin or m1.k != m2.k or m2.dil != 1 or m1.stride != 1 or m2.stride != 1
            or m1.groups != 1 or m2.groups != 1 or m1.transposed or m2.transposed
            or m1.pad != m1.dil * (m1.k - 1) // 2 or m2.pad != (m2.k - 1) // 2 or s1.bank.impl != L.IMPL_AUTO):
        return None
    p = L.ResUnitParams(L.dt_code(x.dtype), x.size(0), x.size(1), m1.cin, m1.k, m1.dil, float(slope))
    return p if L.lib().evt_resunit_wide_supported(C.byref(p)) else None


def _t1_unit(e0, m1, m2, x):
    """trace record of one fused step: flops of both convolutions; bytes = x in, xa / mid_a / y out, both weight images"""
    e0, rf = e0
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    if rf is not None:
        rf.__exit__(None, None, None)
    n, ln, c = x.shape
    macs = n * ln * c * c * (m1.k + m2.k)
    TRACE.append((L.lib().evt_last_kernel_tag().decode(), "fwd", 2 * macs, 4 * x.numel() * 2 + (m1.v.numel() + m2.v.numel()) * 2,
                  e0, e1, f"unit {c}>{c} k{m1.k} d{m1.dil} n{n} L{ln}", m1))


def _lrelu(x, slope, owner=None):
    out = torch.empty_like(x)
    e0 = _t0()
    L.check(L.lib().evt_leaky_relu(L.dt_of(x), L.ptr(x), C.c_float(slope), L.ptr(out), C.c_int64(x.numel()),
                                   L.stream_ptr()), "evt_leaky_relu")
    if e0 is not None:
        _t1_elt(e0, "lrelu_kernel", 2 * x.numel() * x.element_size(), owner)
    return out


def _add3(a, b, c, scale, out, owner):
    e0 = _t0()
    L.check(L.lib().evt_add3_scale(L.dt_of(a), L.ptr(a), L.ptr(b), L.ptr(c), C.c_float(scale), L.ptr(out),
                                   C.c_int64(a.numel()), L.stream_ptr()), "evt_add3_scale")
    if e0 is not None:
        _t1_elt(e0, "add3_scale_kernel", (2 + (b is not None) + (c is not None)) * a.numel() * a.element_size(), owner)
    return out


class ResUnitFn(torch.autograd.Function):
    """HiFi-GAN ResBlock1 inner step  y = x + c2(lrelu(c1(lrelu(x))))  (modules.py:299-308).
    Forward: xa = lrelu(x) (one element-wise launch), mid_a = lrelu(c1(xa)) (activation = c1's epilogue),
    y = c2(mid_a) + x (residual = c2's epilogue).  Every convolution and weight gradient therefore sees PLAIN operands
    and runs on the LDS-DMA kernels; the two leaky-relu derivatives are the gate epilogues of the backward-data
    launches (the sign of lrelu(v) is the sign of v), the residual gradient is the second one's add epilogue."""

    @staticmethod
    def forward(ctx, x, anchor, s1, s2, slope):
        fused = _resunit_params(s1, s2, x, slope)
        if fused is not None:
            # narrow stages (C = 16 / 32): the whole step in one launch (csrc/resunit.hip); it also writes the two
            # activated tensors the backward launches below take
            m1, m2 = s1.module, s2.module
            xa, mid_a, y = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
            e0 = _t0()
            L.check(L.lib().evt_resunit_fwd(C.byref(fused), L.ptr(x), L.ptr(s1.reg), L.ptr(s2.reg),
                                            L.ptr(m1.bias.data if m1.bias is not None else None),
                                            L.ptr(m2.bias.data if m2.bias is not None else None), L.ptr(xa), L.ptr(mid_a),
                                            L.ptr(y), L.stream_ptr()), "evt_resunit_fwd")
            if e0 is not None:
                _t1_unit(e0, m1, m2, x)
            ctx.s1, ctx.s2, ctx.slope = s1, s2, slope
            ctx.save_for_backward(xa, mid_a)
            return y
        wide = _resunit_wide_params(s1, s2, x, slope)
        if wide is not None:
            # wide stages (C = 64 / 128): one launch, the intermediate stays in LDS (csrc/resunit_wide.hip)
            m1, m2 = s1.module, s2.module
            xa, mid_a, y = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
            e0 = _t0()
            L.check(L.lib().evt_resunit_wide_fwd(C.byref(wide), L.ptr(x), L.ptr(s1.reg), L.ptr(s2.reg),
                                                 L.ptr(m1.bias.data if m1.bias is not None else None),
                                                 L.ptr(m2.bias.data if m2.bias is not None else None), L.ptr(xa),
                                                 L.ptr(mid_a), L.ptr(y), L.stream_ptr()), "evt_resunit_wide_fwd")
            if e0 is not None:
                _t1_unit(e0, m1, m2, x)
            ctx.s1, ctx.s2, ctx.slope = s1, s2, slope
            ctx.save_for_backward(xa, mid_a)
            return y
        xa = _lrelu(x, slope, s1.module)
        mid_a = _fwd(s1, xa, None, 1.0, L.ACT_LRELU, slope)
        y = _fwd(s2, mid_a, x, 1.0, L.ACT_NONE, 1.0)
        ctx.s1, ctx.s2, ctx.slope = s1, s2, slope
        ctx.save_for_backward(xa, mid_a)
        return y

    @staticmethod
    def backward(ctx, dy):
        xa, mid_a = ctx.saved_tensors
        s1, s2, slope = ctx.s1, ctx.s2, ctx.slope
        dy = dy.contiguous()
        nseq, lin = xa.size(0), xa.size(1)
        if ctx.needs_input_grad[0]:
            dx = resunit_bwd(s1, s2, dy, xa, mid_a, slope, 1.0)
            if dx is not None:
                return dx, None, None, None, None
            wide = _resunit_wide_params(s1, s2, xa, slope)
            if wide is not None:
                m1, m2 = s1.module, s2.module
                dmid, dx = torch.empty_like(dy), torch.empty_like(dy)
                e0 = _t0()
                L.check(L.lib().evt_resunit_wide_bwd_data(C.byref(wide), L.ptr(dy), C.c_float(1.0), L.ptr(xa), L.ptr(mid_a),
                                                          L.ptr(s1.alt), L.ptr(s2.alt), L.ptr(dmid), L.ptr(dx),
                                                          L.stream_ptr()), "evt_resunit_wide_bwd_data")
                if e0 is not None:
                    _t1_unit_bwd(e0, m1, m2, xa, False)
                if s2.bank.weight_grads:
                    _bwd_weight(s2, mid_a, dy, None, nseq, lin, 1.0, L.ACT_NONE, 1.0)
                    _bwd_weight(s1, xa, dmid, None, nseq, lin, 1.0, L.ACT_NONE, 1.0)
                return dx, None, None, None, None
        if s2.bank.weight_grads:
            _bwd_weight(s2, mid_a, dy, None, nseq, lin, 1.0, L.ACT_NONE, 1.0)
        # d(c1 output before its activation) = (W2^T dy) * lrelu'(mid): gate epilogue on mid_a
        dmid = _bwd_data(s2, dy, None, mid_a, None, nseq, lin, slope, L.ACT_NONE, 1.0)
        if s1.bank.weight_grads:
            _bwd_weight(s1, xa, dmid, None, nseq, lin, 1.0, L.ACT_NONE, 1.0)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _bwd_data(s1, dmid, None, xa, dy, nseq, lin, slope, L.ACT_NONE, 1.0)
        return dx, None, None, None, None


def _t1_unit_bwd(e0, m1, m2, xa, wg):
    """trace record of one fused backward: both backward-data convolutions (+ both weight gradients); bytes = dy, xa,
    mid_a in, dx out, both weight images (+ both fp32 gradient images)"""
    e0, rf = e0
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    if rf is not None:
        rf.__exit__(None, None, None)
    n, ln, c = xa.shape
    macs = n * ln * c * c * (m1.k + m2.k) * (2 if wg else 1)
    wb = (m1.v.numel() + m2.v.numel()) * (2 + (4 if wg else 0))
    TRACE.append((L.lib().evt_last_kernel_tag().decode(), "bwd_unit", 2 * macs, 4 * xa.numel() * 2 + wb, e0, e1,
                  f"unit {c}>{c} k{m1.k} d{m1.dil} n{n} L{ln}", m1))


def resunit_bwd(s1, s2, dy, xa, mid_a, slope, dy_scale=1.0):
    """The whole backward of one ResBlock1 step in one launch (csrc/resunit_bwd.hip) when the shape is covered: returns
    dx (and leaves the weight / bias gradients of both convolutions in their gradient images / .grad), else None.
    Where the kernel takes the data gradients but not the weight gradients (C = 32 with 11 taps: the accumulators do
    not fit the register file) it also writes dmid and the two weight gradients run as their own launches."""
    fused = _resunit_params(s1, s2, xa, slope)
    if fused is None:
        return None
    lib = L.lib()
    bank = s1.bank
    wg = bool(bank.weight_grads)
    wg_in = wg and bool(lib.evt_resunit_bwd_supported(C.byref(fused), 1))
    if not lib.evt_resunit_bwd_supported(C.byref(fused), 0):
        return None
    if wg and not wg_in and dy_scale != 1.0:
        return None
    m1, m2 = s1.module, s2.module
    dx = torch.empty_like(dy)
    dmid = torch.empty_like(dy) if (wg and not wg_in) else None
    dw1 = dw2 = db1 = db2 = ws = None
    if wg_in:
        for m in (m1, m2):
            if m.bias is not None and m.bias.grad is None:
                m.bias.grad = torch.zeros_like(m.bias)
        dw1, dw2 = s1.dw, s2.dw
        db1 = m1.bias.grad if m1.bias is not None else None
        db2 = m2.bias.grad if m2.bias is not None else None
        ws = bank.scratch()
    e0 = _t0()
    L.check(lib.evt_resunit_bwd(C.byref(fused), L.ptr(dy), C.c_float(dy_scale), L.ptr(xa), L.ptr(mid_a), L.ptr(s1.alt),
                                L.ptr(s2.alt), L.ptr(dx), L.ptr(dmid), L.ptr(dw1), L.ptr(dw2), L.ptr(db1), L.ptr(db2),
                                L.ptr(ws), C.c_int64(ws.numel() if ws is not None else 0), L.stream_ptr()),
            "evt_resunit_bwd")
    if e0 is not None:
        _t1_unit_bwd(e0, m1, m2, xa, wg_in)
        if wg_in:
            _t1_elt(_t0(), "fold_partials_multi", (s1.layout.reg_elems + s2.layout.reg_elems) * 4 * 256, m1)
    if wg_in:
        s1.wg_dirty = s2.wg_dirty = True
    elif wg:
        nseq, lin = xa.size(0), xa.size(1)
        _bwd_weight(s2, mid_a, dy, None, nseq, lin, 1.0, L.ACT_NONE, 1.0)
        _bwd_weight(s1, xa, dmid, None, nseq, lin, 1.0, L.ACT_NONE, 1.0)
    return dx


def res_unit(x, c1: EvtConv1d, c2: EvtConv1d, slope: float):
    return ResUnitFn.apply(x, c1._slot.bank.anchor, c1._slot, c2._slot, float(slope))


def _dptr(t):
    return t.data_ptr() if t is not None else None


def _t1_multi(e0, kind, pairs, x, wg):
    """trace record of one grouped launch: the sums over its jobs of what _t1_unit / _t1_unit_bwd record"""
    e0, rf = e0
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    if rf is not None:
        rf.__exit__(None, None, None)
    n, ln, c = x.shape
    macs = sum(n * ln * c * c * (s1.module.k + s2.module.k) for s1, s2 in pairs)
    wel = sum(s1.module.v.numel() + s2.module.v.numel() for s1, s2 in pairs)
    if kind == "bwd_unit":
        macs *= 2 if wg else 1
        wb = wel * (2 + (4 if wg else 0))
    else:
        wb = wel * 2
    m1 = pairs[0][0].module
    TRACE.append((L.lib().evt_last_kernel_tag().decode(), kind, 2 * macs, len(pairs) * 4 * x.numel() * 2 + wb, e0, e1,
                  f"units x{len(pairs)} {c}>{c} k" + "/".join(str(s1.module.k) for s1, _ in pairs) + f" d{m1.dil} n{n} L{ln}", m1))


def _stage_plan(x, blocks, slope):
    """[(slot pairs of unit j over the blocks)] when every step of the stage is covered by the grouped kernels: the blocks
    have kernel sizes 3 / 7 / 11 (one each), equally many steps, and every step passes _resunit_params"""
    if not x.is_cuda or not L.is_half(x.dtype) or len(blocks) != 3:
        return None
    nunits = len(blocks[0].convs1)
    if any(len(b.convs1) != nunits or len(b.convs2) != nunits for b in blocks):
        return None
    if sorted(b.convs1[0].k for b in blocks) != [3, 7, 11]:
        return None
    plan = []
    for j in range(nunits):
        pairs = [(b.convs1[j]._slot, b.convs2[j]._slot) for b in blocks]
        if any(s1 is None or s2 is None for s1, s2 in pairs):
            return None
        ps = [_resunit_params(s1, s2, x, slope) for s1, s2 in pairs]
        if any(p is None for p in ps):
            return None
        plan.append((pairs, ps))
    return plan


class ResStageFn(torch.autograd.Function):
    """One HiFi-GAN stage after its up-sampling convolution (models.py:457-466): the three ResBlock1 of kernel sizes
    3 / 7 / 11 applied to the same input and averaged.  Step j of all three blocks is ONE grouped launch each way
    (csrc/resunit.hip, resunit_bwd.hip), so a stage is 3 + 1 launches forward and 3 + 1 (+ 1 fold each) backward instead
    of 9 + 1 and 36 + 2.  The 1 / 3 of the mean is folded into the first backward launch's load of dy."""

    @staticmethod
    def forward(ctx, x, anchor, plan, slope, scale):
        lib = L.lib()
        cur = [x] * len(plan[0][0])
        saved = []
        for pairs, ps in plan:
            jobs = (L.ResUnitFwdJob * len(pairs))()
            outs = []
            for i, ((s1, s2), p) in enumerate(zip(pairs, ps)):
                m1, m2 = s1.module, s2.module
                xa, mid_a, y = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
                jb = jobs[i]
                jb.p = p
                jb.x, jb.w1_reg, jb.w2_reg = cur[i].data_ptr(), s1.reg.data_ptr(), s2.reg.data_ptr()
                jb.b1 = _dptr(m1.bias.data if m1.bias is not None else None)
                jb.b2 = _dptr(m2.bias.data if m2.bias is not None else None)
                jb.xa, jb.mid_a, jb.y = xa.data_ptr(), mid_a.data_ptr(), y.data_ptr()
                outs.append(y)
                saved += [xa, mid_a]
            e0 = _t0()
            L.check(lib.evt_resunit_fwd_multi(jobs, len(pairs), L.stream_ptr()), "evt_resunit_fwd_multi")
            if e0 is not None:
                _t1_multi(e0, "fwd", pairs, x, False)
            cur = outs
        owner = plan[0][0][0][0].module
        out = _add3(cur[0], cur[1], cur[2], scale, torch.empty_like(x), owner)
        ctx.plan, ctx.slope, ctx.scale = plan, slope, scale
        ctx.save_for_backward(*saved)
        return out

    @staticmethod
    def backward(ctx, dy):
        lib = L.lib()
        plan, slope, scale = ctx.plan, ctx.slope, ctx.scale
        saved = ctx.saved_tensors
        dy = dy.contiguous()
        nb = len(plan[0][0])
        bank = plan[0][0][0][0].bank
        wg = bool(bank.weight_grads)
        # which jobs accumulate their weight gradients in the launch (all but C = 32 with 11 taps)
        in_k = [[wg and bool(lib.evt_resunit_bwd_supported(C.byref(p), 1)) for p in ps] for _, ps in plan]
        fold_scale = all(all(r) for r in in_k) or not wg
        owner = plan[0][0][0][0].module
        if not fold_scale:
            dy = _add3(dy, None, None, scale, torch.empty_like(dy), owner)
        d = [dy] * nb
        ws = bank.scratch() if wg else None
        for j in range(len(plan) - 1, -1, -1):
            pairs, ps = plan[j]
            jobs = (L.ResUnitBwdJob * nb)()
            outs, later = [], []
            for i, ((s1, s2), p) in enumerate(zip(pairs, ps)):
                m1, m2 = s1.module, s2.module
                xa, mid_a = saved[2 * (j * nb + i)], saved[2 * (j * nb + i) + 1]
                dx = torch.empty_like(dy)
                jb = jobs[i]
                jb.p = p
                jb.dy_scale = scale if (fold_scale and j == len(plan) - 1) else 1.0
                jb.dy, jb.xa, jb.mid_a = d[i].data_ptr(), xa.data_ptr(), mid_a.data_ptr()
                jb.w1_alt, jb.w2_alt, jb.dx = s1.alt.data_ptr(), s2.alt.data_ptr(), dx.data_ptr()
                if in_k[j][i]:
                    for m in (m1, m2):
                        if m.bias is not None and m.bias.grad is None:
                            m.bias.grad = torch.zeros_like(m.bias)
                    jb.dw1, jb.dw2 = s1.dw.data_ptr(), s2.dw.data_ptr()
                    jb.db1 = _dptr(m1.bias.grad if m1.bias is not None else None)
                    jb.db2 = _dptr(m2.bias.grad if m2.bias is not None else None)
                    s1.wg_dirty = s2.wg_dirty = True
                elif wg:
                    dmid = torch.empty_like(dy)
                    jb.dmid = dmid.data_ptr()
                    later.append((s1, s2, xa, mid_a, d[i], dmid))
                outs.append(dx)
            e0 = _t0()
            L.check(lib.evt_resunit_bwd_multi(jobs, nb, L.ptr(ws), C.c_int64(ws.numel() if ws is not None else 0),
                                              L.stream_ptr()), "evt_resunit_bwd_multi")
            if e0 is not None:
                _t1_multi(e0, "bwd_unit", pairs, dy, wg)
                if any(in_k[j]):       # the second launch of the call: the partial rows added into the gradient images
                    rows = sum(s1.layout.reg_elems + s2.layout.reg_elems for (s1, s2), k_ in zip(pairs, in_k[j]) if k_)
                    _t1_elt(_t0(), "fold_partials_multi", rows * 4 * 256, owner)
            nseq, lin = dy.size(0), dy.size(1)
            for s1, s2, xa, mid_a, dyi, dmid in later:
                _bwd_weight(s2, mid_a, dyi, None, nseq, lin, 1.0, L.ACT_NONE, 1.0)
                _bwd_weight(s1, xa, dmid, None, nseq, lin, 1.0, L.ACT_NONE, 1.0)
            d = outs
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _add3(d[0], d[1], d[2], 1.0, torch.empty_like(dy), owner)
        return dx, None, None, None, None


def res_stage(x, blocks, slope: float, scale: float):
    """mean over `blocks` (ResBlock1 modules) of block(x), through the grouped kernels; None when the stage is not covered
    (the caller then runs the blocks one by one)"""
    if os.environ.get("EVT_NO_RESSTAGE") == "1":
        return None
    plan = _stage_plan(x, blocks, slope)
    if plan is None:
        return None
    lib = L.lib()
    # the grouped backward exists for every job shape the forward takes (data gradients at least)
    if not all(lib.evt_resunit_bwd_supported(C.byref(p), 0) for _, ps in plan for p in ps):
        return None
    # jobs are handed over in kernel-size order 3 / 7 / 11
    order = sorted(range(len(blocks)), key=lambda i: blocks[i].convs1[0].k)
    plan = [([pairs[i] for i in order], [ps[i] for i in order]) for pairs, ps in plan]
    return ResStageFn.apply(x, plan[0][0][0][0].bank.anchor, plan, float(slope), float(scale))


class Add3ScaleFn(torch.autograd.Function):
    """(a + b + c) * scale — HiFi-GAN stage mean (models.py:457-466)."""

    @staticmethod
    def forward(ctx, a, b, c, scale, owner=None):
        out = _add3(a, b, c, scale, torch.empty_like(a), owner)
        ctx.scale, ctx.owner = scale, owner
        return out

    @staticmethod
    def backward(ctx, d):
        d = d.contiguous()
        g = _add3(d, None, None, ctx.scale, torch.empty_like(d), ctx.owner)
        return g, g, g, None, None


class GatedActFn(torch.autograd.Function):
    """tanh(a+ga) * sigmoid(b+gb) on channels-last [n, t, 2H] (+ g [n, 2H]) — commons.py:94-101."""

    @staticmethod
    def forward(ctx, xin, g):
        n, t, h2 = xin.shape
        acts = torch.empty((n, t, h2 // 2), dtype=xin.dtype, device=xin.device)
        L.check(L.lib().evt_gated_act_fwd(L.dt_of(xin), L.ptr(xin), L.ptr(g), L.ptr(acts), n, t, h2 // 2,
                                          L.stream_ptr()), "evt_gated_act_fwd")
        ctx.save_for_backward(xin, g)
        return acts

    @staticmethod
    def backward(ctx, dacts):
        xin, g = ctx.saved_tensors
        n, t, h2 = xin.shape
        dacts = dacts.contiguous()
        dxin = torch.empty_like(xin)
        dg32 = None
        if g is not None and ctx.needs_input_grad[1]:
            dg32 = torch.zeros((n, h2), dtype=torch.float32, device=xin.device)
        L.check(L.lib().evt_gated_act_bwd(L.dt_of(xin), L.ptr(xin), L.ptr(g), L.ptr(dacts), L.ptr(dxin), L.ptr(dg32),
                                          n, t, h2 // 2, L.stream_ptr()), "evt_gated_act_bwd")
        return dxin, (dg32.to(g.dtype) if dg32 is not None else None)
