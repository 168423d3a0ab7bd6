"""ctypes binding of libevt_hip.so (the C ABI declared in include/evt.h).

The product path has NO fallback: if the shared library is missing or a call returns non-zero, an
exception is raised.  torch is used only to own device memory and streams; every pointer crossing
this boundary is a raw `data_ptr()`.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "libevt_hip.so")           # bfloat16 build
LIB_PATH_F16 = os.path.join(os.path.dirname(_HERE), "libevt_hip_f16.so")   # the same sources with IEEE half (fp16_run)

DT_F32, DT_BF16, DT_F16 = 0, 1, 2
HALF_DTYPES = (torch.bfloat16, torch.float16)
ACT_NONE, ACT_LRELU, ACT_TANH = 0, 1, 2
IMPL_AUTO, IMPL_NAIVE, IMPL_IGEMM = 0, 1, 2


class EvtError(RuntimeError):
    pass


class ConvParams(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("nseq", C.c_int32), ("lin", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32),
        ("k", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("dil", C.c_int32), ("groups", C.c_int32),
        ("transposed", C.c_int32), ("in_slope", C.c_float), ("out_act", C.c_int32), ("out_slope", C.c_float),
        ("impl", C.c_int32),
    ]


class ResUnitParams(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("nseq", C.c_int32), ("L", C.c_int32), ("C", C.c_int32), ("k", C.c_int32),
                ("dil", C.c_int32), ("slope", C.c_float)]


class ResUnitFwdJob(C.Structure):
    _fields_ = [("p", ResUnitParams), ("x", C.c_void_p), ("w1_reg", C.c_void_p), ("w2_reg", C.c_void_p),
                ("b1", C.c_void_p), ("b2", C.c_void_p), ("xa", C.c_void_p), ("mid_a", C.c_void_p), ("y", C.c_void_p)]


class ResUnitBwdJob(C.Structure):
    _fields_ = [("p", ResUnitParams), ("dy_scale", C.c_float), ("dy", C.c_void_p), ("xa", C.c_void_p),
                ("mid_a", C.c_void_p), ("w1_alt", C.c_void_p), ("w2_alt", C.c_void_p), ("dx", C.c_void_p),
                ("dmid", C.c_void_p), ("dw1", C.c_void_p), ("dw2", C.c_void_p), ("db1", C.c_void_p), ("db2", C.c_void_p)]


class WLayout(C.Structure):
    _fields_ = [
        ("d0", C.c_int32), ("d1", C.c_int32), ("k", C.c_int32), ("stride", C.c_int32),
        ("reg_ck", C.c_int32), ("reg_nchunk", C.c_int32), ("reg_kp", C.c_int32),
        ("alt_ck", C.c_int32), ("alt_nchunk", C.c_int32), ("alt_kp", C.c_int32), ("alt_nphase", C.c_int32),
        ("reg_elems", C.c_int64), ("alt_elems", C.c_int64),
    ]


class WPrepItem(C.Structure):
    _fields_ = [
        ("v", C.c_void_p), ("g", C.c_void_p), ("reg", C.c_void_p), ("alt", C.c_void_p),
        ("dw", C.c_void_p), ("dv", C.c_void_p), ("dg", C.c_void_p),
        ("lay", WLayout), ("dtype", C.c_int32), ("src_d1", C.c_int32),
        ("dw_extra", C.c_void_p), ("dw_part_stride", C.c_int64), ("db_part", C.c_void_p), ("db", C.c_void_p),
        ("used", C.c_void_p),
    ]


class FragItem(C.Structure):
    """evt_frag_item (include/evt.h): one REG image -> fragment order copy of evt_frag_pack"""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("rows", C.c_int32), ("ktot", C.c_int32)]


class WgradParts(C.Structure):
    _fields_ = [("dw_extra", C.c_void_p), ("part_stride", C.c_int64), ("db_part", C.c_void_p), ("used_dev", C.c_void_p),
                ("parts", C.c_int32), ("prev_used", C.c_int32), ("used", C.c_int32), ("dirty0", C.c_int32),
                ("ws", C.c_void_p), ("ws_floats", C.c_int64)]


class Seg(C.Structure):
    _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("da", C.c_void_p), ("n", C.c_int64),
                ("scale", C.c_float), ("pad_", C.c_int32)]


class KlParams(C.Structure):
    _fields_ = [("B", C.c_int32), ("T", C.c_int32), ("C", C.c_int32), ("time_inner", C.c_int32),
                ("dt_z_p", C.c_int32), ("dt_logs_q", C.c_int32), ("dt_m_p", C.c_int32), ("dt_logs_p", C.c_int32)]


class AdamWSeg(C.Structure):
    _fields_ = [("begin", C.c_int64), ("end", C.c_int64), ("lr", C.c_float), ("weight_decay", C.c_float)]


class AttnParams(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("B", C.c_int32), ("L", C.c_int32), ("H", C.c_int32), ("D", C.c_int32),
        ("x_len", C.c_int32),
        ("q_stride_b", C.c_int64), ("q_stride_l", C.c_int64), ("q_stride_h", C.c_int64),
        ("o_stride_b", C.c_int64), ("o_stride_l", C.c_int64), ("o_stride_h", C.c_int64),
        ("dropout_p", C.c_float), ("seed", C.c_uint32),
    ]


class SampleParams(C.Structure):
    _fields_ = [("V", C.c_int32), ("eos", C.c_int32), ("top_k", C.c_int32), ("no_eos_steps", C.c_int32),
                ("ymax", C.c_int32), ("top_p", C.c_float), ("temperature", C.c_float),
                ("repetition_penalty", C.c_float), ("seed", C.c_uint32), ("noise_rows", C.c_int32)]


DEC_POS, DEC_IDX, DEC_YCOUNT, DEC_YLEN, DEC_SEED = 0, 1, 2, 3, 4


class SAChunk(C.Structure):
    _fields_ = [("begin", C.c_int64), ("end", C.c_int64), ("tensor", C.c_int32), ("pad_", C.c_int32)]


class ScaledAdamHP(C.Structure):
    _fields_ = [("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("scalar_lr_scale", C.c_float), ("scalar_max", C.c_float), ("step", C.c_int32), ("pad_", C.c_int32)]


class MhaParams(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("B", C.c_int32), ("H", C.c_int32), ("D", C.c_int32), ("Tq", C.c_int32),
                ("Tk", C.c_int32), ("window", C.c_int32), ("n_heads_rel", C.c_int32), ("ldq", C.c_int64),
                ("ldk", C.c_int64), ("ldo", C.c_int64), ("scale", C.c_float), ("dropout_p", C.c_float),
                ("site", C.c_uint32), ("pad_", C.c_uint32), ("seed_dev", C.c_void_p)]


_libs = {}                   # half dtype -> loaded library
_half = torch.bfloat16       # the 16-bit type of the library lib() returns


def set_half(dtype) -> None:
    """Select which build lib() returns: torch.bfloat16 -> libevt_hip.so, torch.float16 -> libevt_hip_f16.so (float32
    leaves the selection alone: both builds serve it).  The engines call this at the top of every step with their compute
    dtype; a process that only ever uses one 16-bit type never notices."""
    global _half
    if dtype in HALF_DTYPES:
        _half = dtype


def half():
    return _half


def is_half(dtype) -> bool:
    return dtype in HALF_DTYPES


def dt_code(dtype) -> int:
    """torch dtype -> EVT_DT_* of the ACTIVE library; a 16-bit type the active build does not serve is an error here, before
    any pointer reaches a kernel that would read its bits as the other format"""
    if dtype == torch.float32:
        return DT_F32
    if dtype in HALF_DTYPES:
        if dtype != _half:
            raise EvtError(f"{dtype} data while the {_half} build of the library is selected: call hip.lib.set_half({dtype}) "
                           "(the engines do, at the top of a step)")
        return DT_BF16 if dtype == torch.bfloat16 else DT_F16
    raise EvtError(f"unsupported dtype {dtype}")


def _check_fresh(handle, LIB_PATH=LIB_PATH):
    """the library carries the hash of the sources it was built from (evt_version(): "... src=<hash>"); next to a source
    tree (this repository, the GPU box's copy of it) a different hash means a stale .so -- an error, not an old kernel"""
    csrc = os.path.join(os.path.dirname(_HERE), "csrc")
    if os.environ.get("EVT_ALLOW_STALE_LIB") == "1" or not os.path.isdir(csrc):
        return
    from ..build import source_hash

    have = handle.evt_version().decode()
    want = source_hash()
    if f"src={want}" not in have:
        raise EvtError(f"{LIB_PATH} is stale: built from sources {have.split('src=')[-1]!r}, the tree has {want!r}; "
                       "rebuild with `python -m easevoice_trainer_amd.build` (EVT_ALLOW_STALE_LIB=1 overrides)")


def lib():
    """The selected build (set_half), loaded once; raise loudly if it is missing (no CPU / eager fallback)."""
    _lib = _libs.get(_half)
    if _lib is None:
        path = LIB_PATH if _half == torch.bfloat16 else LIB_PATH_F16
        if not os.path.exists(path):
            raise EvtError(
                f"{path} not found: build it with `python -m easevoice_trainer_amd.build` "
                "(or __graft_entry__.build()). There is no fallback path.")
        _lib = C.CDLL(path)
        _lib.evt_version.restype = C.c_char_p
        _check_fresh(_lib, path)
        _lib.evt_half_dtype.restype = C.c_int32
        if _lib.evt_half_dtype() != (DT_BF16 if _half == torch.bfloat16 else DT_F16):
            raise EvtError(f"{path} serves half code {_lib.evt_half_dtype()}: not the {_half} build")
        _libs[_half] = _lib
        _lib.evt_conv1d_lout.restype = C.c_int32
        _lib.evt_mel_workspace_floats.restype = C.c_int64
        _lib.evt_workspace_bytes.restype = C.c_int64
        _lib.evt_last_kernel_tag.restype = C.c_char_p
        _lib.evt_resunit_bwd_ws_floats.restype = C.c_int64
        _lib.evt_debug_kernel_tags.restype = None
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise EvtError(f"{what} failed with code {rc}")


def dt_of(t: torch.Tensor) -> int:
    return dt_code(t.dtype)


def torch_dtype(dt: int) -> torch.dtype:
    return {DT_F32: torch.float32, DT_BF16: torch.bfloat16, DT_F16: torch.float16}[dt]


def ptr(t):
    """device pointer of a tensor (None -> NULL); the tensor must be contiguous"""
    if t is None:
        return C.c_void_p(0)
    if not t.is_contiguous():
        raise EvtError("non-contiguous tensor passed to the C ABI")
    return C.c_void_p(t.data_ptr())


_hip_rt = None
_role_pools = {}


def own_stream(device):
    """a HIP stream created for the library (hipStreamCreateWithFlags, non-blocking), wrapped for torch.  torch.cuda.Stream()
    hands out one of 32 pooled streams per device round-robin: in a process that has asked for more than that (every engine
    takes a few; a test run takes hundreds) two "different" side streams -- or a side stream and torch's graph-capture
    stream -- are the same HIP stream, and a capture that forks onto such a pair has ended in a segmentation fault inside
    hipStreamEndCapture (round 6: tests in the order streams, book-pipe, graph; profiles/r06_streams.txt).  Never destroyed:
    callers take theirs through role_stream(), which bounds the number."""
    global _hip_rt
    device = torch.device(device)
    if _hip_rt is None:
        # the HIP runtime this process already runs on (torch's): opened by the path it is mapped from, so that dlopen can
        # only hand back that instance -- a second copy of the runtime would create streams the first does not know
        path = "libamdhip64.so"
        try:
            with open("/proc/self/maps") as f:
                for line in f:
                    if "libamdhip64.so" in line:
                        path = line.split(None, 5)[-1].strip()
                        break
        except OSError:
            pass
        _hip_rt = C.CDLL(path)
        _hip_rt.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
        _hip_rt.hipStreamCreateWithFlags.restype = C.c_int
    h = C.c_void_p()
    with torch.cuda.device(device):
        rc = _hip_rt.hipStreamCreateWithFlags(C.byref(h), 1)          # hipStreamNonBlocking
    if rc != 0 or not h.value:
        raise EvtError(f"hipStreamCreateWithFlags failed ({rc})")
    return torch.cuda.ExternalStream(h.value, device=device)


def role_stream(device, role: str, ring: int = 4):
    """the next of `ring` library-owned streams kept per (device, role) -- "bank" (weight-gradient side streams), "book",
    "lane0", "lane1", "comm" ...: streams of different roles are never the same HIP stream, streams of one role repeat
    after `ring` requests (two banks of one engine get two; the engine after the next one re-uses the first pair)"""
    device = torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    if os.environ.get("EVT_POOL_STREAMS", "0") == "1":               # A/B switch: torch's pooled streams, as before
        return torch.cuda.Stream(device=device)
    pool = _role_pools.setdefault((device.index, role), [[], 0])
    if len(pool[0]) < ring:
        pool[0].append(own_stream(device))
        return pool[0][-1]
    st = pool[0][pool[1] % ring]
    pool[1] += 1
    return st


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_PINNED_KEEP = []   # host staging buffers referenced by captured H2D copy nodes must outlive the graphs


def struct_to_device(items, device) -> torch.Tensor:
    """copy a python list of ctypes structures to a device uint8 tensor.  The staging buffer is pinned and the copy
    is asynchronous on the current stream, so the call is legal while a HIP graph is being captured (the copy becomes
    a graph node that re-reads the pinned buffer on every replay; the buffer is kept alive for that)."""
    if not items:
        raise EvtError("empty table")
    arr = (type(items[0]) * len(items))(*items)
    buf = bytes(arr)
    device = torch.device(device)
    if device.type != "cuda":
        return torch.frombuffer(bytearray(buf), dtype=torch.uint8).to(device)
    host = torch.empty(len(buf), dtype=torch.uint8, pin_memory=True)
    host.copy_(torch.frombuffer(bytearray(buf), dtype=torch.uint8))
    dev = torch.empty(len(buf), dtype=torch.uint8, device=device)
    dev.copy_(host, non_blocking=True)
    if torch.cuda.is_current_stream_capturing():
        _PINNED_KEEP.append(host)
    else:
        dev._evt_host = host     # the copy may still be in flight when this function returns
    return dev
