"""Minimal TensorBoard event-file writer/reader for scalar summaries (no TensorFlow / tensorboard dependency).

File format (tensorflow/core/lib/io/record_writer.cc, tensorflow/core/util/event.proto, summary.proto):
  record  = uint64 length | uint32 masked_crc32c(length bytes) | data | uint32 masked_crc32c(data)
  Event   = { 1: double wall_time, 2: int64 step, 3: string file_version | 5: Summary summary }
  Summary = { 1: repeated Value { 1: string tag, 2: float simple_value } }
The reference logs its scalars with tf.summary.FileWriter (paac.py:152-155, actor_learner.py:83,
policy_monitor.py:111-115); files written here open in TensorBoard the same way."""
import os
import socket
import struct
import time

_CRC_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ 0x82F63B78 if _c & 1 else _c >> 1
    _CRC_TABLE.append(_c)


def crc32c(data):
    c = 0xFFFFFFFF
    for b in data:
        c = _CRC_TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n):
    out = bytearray()
    n &= (1 << 64) - 1
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _bytes_field(num, payload):
    return _varint((num << 3) | 2) + _varint(len(payload)) + payload


def encode_scalar_event(tag, value, step, wall_time):
    val = _bytes_field(1, tag.encode()) + _varint((2 << 3) | 5) + struct.pack("<f", float(value))
    summary = _bytes_field(1, val)
    return _varint((1 << 3) | 1) + struct.pack("<d", wall_time) + _varint((2 << 3) | 0) + _varint(int(step)) + _bytes_field(5, summary)


def encode_version_event(wall_time):
    return _varint((1 << 3) | 1) + struct.pack("<d", wall_time) + _bytes_field(3, b"brain.Event:2")


def frame(data):
    head = struct.pack("<Q", len(data))
    return head + struct.pack("<I", masked_crc32c(head)) + data + struct.pack("<I", masked_crc32c(data))


class EventFileWriter(object):
    def __init__(self, logdir):
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, "events.out.tfevents.%010d.%s" % (int(time.time()), socket.gethostname()))
        self._f = open(self.path, "ab")
        self._f.write(frame(encode_version_event(time.time())))

    def add_scalar(self, tag, value, step, wall_time=None):
        self._f.write(frame(encode_scalar_event(tag, value, step, time.time() if wall_time is None else wall_time)))

    def flush(self):
        self._f.flush()

    def close(self):
        self._f.close()


def _read_varint(buf, i):
    n, shift = 0, 0
    while True:
        b = buf[i]
        i += 1
        n |= (b & 0x7F) << shift
        shift += 7
        if not b & 0x80:
            return n, i


def _fields(buf):
    i = 0
    while i < len(buf):
        key, i = _read_varint(buf, i)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _read_varint(buf, i)
        elif wt == 1:
            v, i = buf[i:i + 8], i + 8
        elif wt == 5:
            v, i = buf[i:i + 4], i + 4
        elif wt == 2:
            ln, i = _read_varint(buf, i)
            v, i = buf[i:i + ln], i + ln
        else:
            raise ValueError("unsupported wire type %d" % wt)
        yield num, wt, v


def read_scalars(path):
    """[(tag, value, step, wall_time)] of an event file; checks every record's CRCs."""
    out = []
    with open(path, "rb") as f:
        data = f.read()
    i = 0
    while i < len(data):
        head = data[i:i + 8]
        (ln,) = struct.unpack("<Q", head)
        if struct.unpack("<I", data[i + 8:i + 12])[0] != masked_crc32c(head):
            raise ValueError("corrupt length crc at %d" % i)
        rec = data[i + 12:i + 12 + ln]
        if struct.unpack("<I", data[i + 12 + ln:i + 16 + ln])[0] != masked_crc32c(rec):
            raise ValueError("corrupt data crc at %d" % i)
        i += 16 + ln
        wall, step, summary = 0.0, 0, None
        for num, wt, v in _fields(rec):
            if num == 1 and wt == 1:
                (wall,) = struct.unpack("<d", v)
            elif num == 2 and wt == 0:
                step = v
            elif num == 5 and wt == 2:
                summary = v
        if summary is None:
            continue
        for num, wt, v in _fields(summary):
            if num == 1 and wt == 2:
                tag, val = None, None
                for n2, w2, v2 in _fields(v):
                    if n2 == 1 and w2 == 2:
                        tag = bytes(v2).decode()
                    elif n2 == 2 and w2 == 5:
                        (val,) = struct.unpack("<f", v2)
                out.append((tag, val, step, wall))
    return out
