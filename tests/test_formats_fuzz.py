"""The audio / feature file reader on damaged files: whatever the bytes, sgz_read_audio_file answers with a frame
count the file can hold or with an error -- no crash, no read past the file's end, no allocation sized by a header
field alone.  Valid files of every format the reader knows (float files by the library's writers, 16-bit files by
Python's wave module and by hand) are truncated and have bytes overwritten at random.  CPU only."""
import ctypes as C
import os
import struct
import wave

import numpy as np
import pytest

from strugatzki_amd import _native as N

lib = N.lib
KINDS = ("feature", "correlate", "segmentation", "selfsimilarity")


def _read(path):
    n, ch, sr = C.c_int64(-1), C.c_int(-1), C.c_double(0.0)
    rc = lib.sgz_read_audio_file(path.encode(), None, 0, C.byref(n), C.byref(ch), C.byref(sr))
    if rc != N.SGZ_OK:
        return rc, None
    assert n.value >= 0 and ch.value >= 1
    out = np.empty((n.value, ch.value), dtype=np.float32)
    rc2 = lib.sgz_read_audio_file(path.encode(), N.ptr(out), n.value, None, None, None)
    assert rc2 == N.SGZ_OK
    return rc, out


def _valid_files(tmp_path):
    rng = np.random.default_rng(5)
    x = rng.standard_normal((40, 3)).astype(np.float32)
    files = []
    for kind in ("aiff", "wave", "next", "ircam"):
        p = str(tmp_path / f"f.{kind}")
        assert lib.sgz_write_float_audio(p.encode(), kind.encode(), N.ptr(x), x.shape[0], x.shape[1], 44100.0) == 0
        files.append(p)
    p = str(tmp_path / "pcm16.wav")
    with wave.open(p, "wb") as w:
        w.setnchannels(2)
        w.setsampwidth(2)
        w.setframerate(22050)
        w.writeframes((rng.integers(-3000, 3000, size=(50, 2))).astype("<i2").tobytes())
    files.append(p)
    # a plain 16-bit AIFF (no compression field), written chunk by chunk
    pcm = rng.integers(-3000, 3000, size=(30, 2)).astype(">i2").tobytes()
    rate80 = bytes([0x40, 0x0E, 0xAC, 0x44, 0, 0, 0, 0, 0, 0])  # 44100 as 80-bit extended
    comm = struct.pack(">hIh", 2, 30, 16) + rate80
    ssnd = struct.pack(">II", 0, 0) + pcm
    body = b"AIFF" + b"COMM" + struct.pack(">I", len(comm)) + comm + b"SSND" + struct.pack(">I", len(ssnd)) + ssnd
    p = str(tmp_path / "pcm16.aif")
    open(p, "wb").write(b"FORM" + struct.pack(">I", len(body)) + body)
    files.append(p)
    return files


def test_valid_files_read_back(tmp_path):
    for p in _valid_files(tmp_path):
        rc, out = _read(p)
        assert rc == N.SGZ_OK and out.shape[0] in (30, 40, 50) and np.isfinite(out).all()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_damaged_files_are_refused_or_read_within_their_bytes(tmp_path, seed):
    rng = np.random.default_rng(seed)
    m = str(tmp_path / "mutant")
    ok = bad = 0
    for p in _valid_files(tmp_path):
        good = open(p, "rb").read()
        for trial in range(250):
            d = bytearray(good)
            kind = trial % 5
            if kind == 0:  # cut anywhere
                d = d[:int(rng.integers(0, len(d)))]
            elif kind == 1:  # a few bytes overwritten in the header region
                for _ in range(int(rng.integers(1, 6))):
                    d[int(rng.integers(0, min(len(d), 64)))] = int(rng.integers(0, 256))
            elif kind == 2:  # a 32-bit field set to an extreme value
                at = 4 * int(rng.integers(0, min(len(d), 64) // 4))
                d[at:at + 4] = [b"\xff\xff\xff\xff", b"\x7f\xff\xff\xff", b"\x00\x00\x00\x00", b"\x80\x00\x00\x00"][int(rng.integers(0, 4))]
            elif kind == 3:  # bytes overwritten anywhere, then cut
                for _ in range(int(rng.integers(1, 10))):
                    d[int(rng.integers(0, len(d)))] = int(rng.integers(0, 256))
                d = d[:int(rng.integers(12, len(d) + 1))]
            else:  # garbage appended, a header byte changed
                d[int(rng.integers(0, min(len(d), 48)))] ^= 1 << int(rng.integers(0, 8))
                d += bytes(rng.integers(0, 256, size=int(rng.integers(0, 40))).astype(np.uint8))
            open(m, "wb").write(bytes(d))
            rc, out = _read(m)
            if rc == N.SGZ_OK:
                ok += 1
                assert out.size <= len(d)  # every sample takes a byte of the file at least
            else:
                bad += 1
                assert rc in (N.SGZ_E_IO, N.SGZ_E_ARG)
    assert ok > 0 and bad > 0


@pytest.mark.parametrize("seed", [11, 12])
def test_damaged_config_documents_are_refused_or_parsed(seed):
    """The XML configurations (FeatureCorrelation.scala:231-244 and the other Config.fromXML) with characters
    replaced, dropped or cut off: an error or a document, never a crash."""
    from strugatzki_amd import processors as P
    rng = np.random.default_rng(seed)
    alphabet = "<>/\"'& =ab01-.e\n\t;#x"
    ok = bad = 0
    for kind in KINDS:
        good = P.default_config_xml(kind)
        assert P.roundtrip_native(kind, good)
        for trial in range(300):
            s = list(good)
            what = trial % 4
            if what == 0:
                s = s[:int(rng.integers(0, len(s)))]
            elif what == 1:
                for _ in range(int(rng.integers(1, 5))):
                    s[int(rng.integers(0, len(s)))] = alphabet[int(rng.integers(0, len(alphabet) - 1))]
            elif what == 2:
                a = int(rng.integers(0, len(s)))
                del s[a:a + int(rng.integers(1, 12))]
            else:
                a = int(rng.integers(0, len(s)))
                s[a:a] = list("".join(alphabet[int(i)] for i in rng.integers(0, len(alphabet) - 1, size=int(rng.integers(1, 20)))))
            try:
                P.roundtrip_native(kind, "".join(s))
                ok += 1
            except N.NativeError:
                bad += 1
    assert bad > 0
