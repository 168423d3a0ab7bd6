"""Scalar summaries in TensorBoard's event-file format, written without the tensorboard package (it is an optional
dependency of the reference; the trainers there log through torch.utils.tensorboard.SummaryWriter, src/train/sovits.py:220,
561-565, and Lightning's TensorBoardLogger, src/train/gpt.py:143).  Only what those calls need for curves: `Event` records
with `simple_value` summaries, framed as TFRecords (length, masked CRC-32C, payload, masked CRC-32C).

    Event   { 1: double wall_time, 2: int64 step, 3: string file_version | 5: Summary summary }
    Summary { 1: repeated Value value }      Value { 1: string tag, 2: float simple_value }"""
import os
import socket
import struct
import time

_POLY = 0x82F63B78
_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ _POLY if _c & 1 else _c >> 1
    _TABLE.append(_c)


def crc32c(data: bytes) -> int:
    c = 0xFFFFFFFF
    for b in data:
        c = _TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _masked(data: bytes) -> int:
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n: int) -> bytes:
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _field(num: int, wire: int, payload: bytes) -> bytes:
    head = _varint((num << 3) | wire)
    return head + (_varint(len(payload)) + payload if wire == 2 else payload)


def encode_event(wall_time: float, step: int, scalars=None, file_version=None) -> bytes:
    ev = _field(1, 1, struct.pack("<d", wall_time)) + _field(2, 0, _varint(step))
    if file_version is not None:
        ev += _field(3, 2, file_version.encode())
    if scalars:
        summary = b"".join(_field(1, 2, _field(1, 2, tag.encode()) + _field(2, 5, struct.pack("<f", float(v))))
                           for tag, v in scalars.items())
        ev += _field(5, 2, summary)
    return ev


def frame(record: bytes) -> bytes:
    head = struct.pack("<Q", len(record))
    return head + struct.pack("<I", _masked(head)) + record + struct.pack("<I", _masked(record))


class ScalarWriter:
    """add_scalars(step, {"loss/g/total": 1.2, ...}) -> <log_dir>/events.out.tfevents.<time>.<host>.<pid>"""

    def __init__(self, log_dir: str):
        os.makedirs(log_dir, exist_ok=True)
        self.path = os.path.join(log_dir, "events.out.tfevents.%010d.%s.%d" % (int(time.time()), socket.gethostname(),
                                                                              os.getpid()))
        self._f = open(self.path, "ab")
        self._f.write(frame(encode_event(time.time(), 0, file_version="brain.Event:2")))
        self._f.flush()

    def add_scalars(self, step: int, scalars: dict):
        self._f.write(frame(encode_event(time.time(), int(step), scalars=scalars)))
        self._f.flush()

    def close(self):
        self._f.close()


def tensorboard_log_dir(name=None) -> str:
    """<base>/tb_logs[/<name>] of src/service/tensorboard.py:8-24; <base> is the reference checkout the service runs in
    (the trainer's working directory), or $EVT_TB_DIR"""
    root = os.environ.get("EVT_TB_DIR") or os.path.join(os.getcwd(), "tb_logs")
    return root if name is None else os.path.join(root, name)


def open_writer(log_dir: str):
    """a ScalarWriter, or None when the directory cannot be written: summaries never take a training run down"""
    try:
        return ScalarWriter(log_dir)
    except OSError:
        return None


def log_scalars(writer, step: int, scalars: dict):
    if writer is None:
        return
    try:
        writer.add_scalars(step, scalars)
    except (OSError, ValueError, TypeError):
        pass
