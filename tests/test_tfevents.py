"""TensorBoard event files written without TensorFlow: framing CRCs, protobuf wire format, round trip."""
import struct
import sys, os

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "golds-rl-gym_amd"))
from goldsrl import utils_tfevents as T


def test_crc32c_known_answers():
    # RFC 3720 B.4 test vectors for CRC-32C (Castagnoli)
    assert T.crc32c(b"") == 0
    assert T.crc32c(bytes(32)) == 0x8A9136AA
    assert T.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    assert T.crc32c(bytes(range(32))) == 0x46DD794E
    assert T.crc32c(b"123456789") == 0xE3069283


def test_event_file_round_trip(tmp_path):
    w = T.EventFileWriter(str(tmp_path))
    w.add_scalar("eval/total_reward", -12.5, 1000, wall_time=123.25)
    w.add_scalar("rl/reward", 0.75, 2 ** 40 + 3)
    w.close()
    got = T.read_scalars(w.path)
    assert [(t, v, s) for t, v, s, _ in got] == [("eval/total_reward", -12.5, 1000), ("rl/reward", 0.75, 2 ** 40 + 3)]
    assert got[0][3] == 123.25
    raw = open(w.path, "rb").read()
    (ln,) = struct.unpack("<Q", raw[:8])
    assert raw[12:12 + ln].endswith(b"brain.Event:2")          # first record: file_version
    # a flipped payload byte is caught by the record CRC
    bad = bytearray(raw); bad[-6] ^= 0x40
    p2 = tmp_path / "bad"; p2.write_bytes(bytes(bad))
    try:
        T.read_scalars(str(p2))
        assert False, "corruption not detected"
    except ValueError:
        pass
