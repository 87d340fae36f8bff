"""On-disk formats and command-line conversions, checked against readers that are not the library's own:
a struct-level AIFF-C parser written from the format's specification, scipy's WAVE reader, and the reference's
formulas for the seconds-based options (Strugatzki.scala:143, 145-176).  CPU only."""
import ctypes as C
import os
import struct
import xml.etree.ElementTree as ET

import numpy as np
import pytest

from strugatzki_amd import _native as N
from strugatzki_amd import processors as P

lib = N.lib


def _ext80(b: bytes) -> float:
    """80-bit IEEE extended, big endian: sign / 15-bit exponent (bias 16383) / 64-bit mantissa with explicit 1."""
    exp = ((b[0] & 0x7F) << 8) | b[1]
    mant = int.from_bytes(b[2:10], "big")
    v = mant * 2.0 ** (exp - 16383 - 63) if (exp or mant) else 0.0
    return -v if b[0] & 0x80 else v


def _parse_aifc(path):
    """AIFF-C as Apple's specification lays it out: FORM <size> AIFC, then chunks <id> <size> <data> padded to even
    sizes; FVER carries the version timestamp, COMM channels / frames / sample size / rate (80-bit) / compression
    id / Pascal-string name (padded to even total length), SSND offset / block size / samples."""
    d = open(path, "rb").read()
    assert d[:4] == b"FORM" and d[8:12] == b"AIFC"
    assert struct.unpack(">I", d[4:8])[0] == len(d) - 8  # the FORM size covers the rest of the file
    chunks, p = {}, 12
    while p < len(d):
        cid, n = d[p:p + 4], struct.unpack(">I", d[p + 4:p + 8])[0]
        chunks[cid] = d[p + 8:p + 8 + n]
        p += 8 + n + (n & 1)
    assert p == len(d)
    assert struct.unpack(">I", chunks[b"FVER"])[0] == 0xA2805140
    comm = chunks[b"COMM"]
    nch, frames, bits = struct.unpack(">hIh", comm[:8])
    rate = _ext80(comm[8:18])
    comp = comm[18:22]
    plen = comm[22]
    assert len(comm) == 23 + plen + ((plen + 1) & 1)  # the Pascal string takes an even number of bytes
    ssnd = chunks[b"SSND"]
    off, blk = struct.unpack(">II", ssnd[:8])
    assert blk == 0
    data = np.frombuffer(ssnd[8 + off:], dtype=">f4").astype(np.float32)
    return nch, frames, bits, rate, comp, data.reshape(frames, nch)


@pytest.mark.parametrize("shape,rate", [((7, 14), 44100.0 / 512), ((1, 1), 44100.0), ((33, 3), 48000.0), ((0, 2), 86.1328125)])
def test_float_aiff_is_spec_conformant_aifc_fl32(tmp_path, shape, rate):
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal(shape).astype(np.float32)
    if x.size:
        x.flat[0] = np.float32(-0.0)
    path = str(tmp_path / "f.aif")
    P.write_float_aiff(path, x, rate)
    nch, frames, bits, got_rate, comp, data = _parse_aifc(path)
    assert (nch, frames, bits, comp) == (shape[1], shape[0], 32, b"fl32")
    assert got_rate == rate  # both rates are exact in the 64-bit mantissa of the 80-bit format
    assert np.array_equal(data.view(np.uint32), x.view(np.uint32))  # bit for bit, big endian on disk
    back, r2 = P.read_audio_file(path)
    assert r2 == rate and np.array_equal(back.view(np.uint32).reshape(shape), x.view(np.uint32))


def test_float_wave_and_next_writers_against_independent_readers(tmp_path):
    from scipy.io import wavfile
    rng = np.random.default_rng(4)
    x = rng.standard_normal((50, 2)).astype(np.float32)
    for typ, name in ((b"wav", "o.wav"), (b"next", "o.au")):
        path = str(tmp_path / name)
        assert lib.sgz_write_float_audio(path.encode(), typ, N.ptr(x), x.shape[0], x.shape[1], 86.0) == 0
        back, rate = P.read_audio_file(path)
        assert rate == 86.0 and np.array_equal(back, x)
    rate, data = wavfile.read(str(tmp_path / "o.wav"))
    assert rate == 86 and data.dtype == np.float32 and np.array_equal(data, x)
    d = open(str(tmp_path / "o.au"), "rb").read()
    magic, off, size, enc, rate, ch = struct.unpack(">4sIIIII", d[:24])
    assert (magic, enc, rate, ch, size) == (b".snd", 6, 86, 2, x.size * 4)
    assert np.array_equal(np.frombuffer(d[off:], dtype=">f4").astype(np.float32).reshape(x.shape), x)
    # IRCAM / BICSF: 1024-byte header (magic, fp32 rate, channels, encoding 4 = fp32), big-endian floats behind it
    path = str(tmp_path / "o.sf")
    assert lib.sgz_write_float_audio(path.encode(), b"ircam", N.ptr(x), x.shape[0], x.shape[1], 86.1328125) == 0
    d = open(path, "rb").read()
    assert d[:4] == bytes([0x64, 0xA3, 0x02, 0x00]) and len(d) == 1024 + x.size * 4
    rate, ch, enc = struct.unpack(">fII", d[4:16])
    assert (rate, ch, enc) == (86.1328125, 2, 4) and not any(d[16:1024])
    assert np.array_equal(np.frombuffer(d[1024:], dtype=">f4").astype(np.float32).reshape(x.shape), x)
    back, rate = P.read_audio_file(path)
    assert rate == 86.1328125 and np.array_equal(back, x)
    # ... and a little-endian one as a VAX-flavoured writer leaves it, 16-bit integers
    y = (rng.integers(-30000, 30000, size=(40, 1))).astype("<i2")
    le = bytes([0x64, 0xA3, 0x01, 0x00]) + struct.pack("<fII", 44100.0, 1, 2) + bytes(1008) + y.tobytes()
    open(str(tmp_path / "le.sf"), "wb").write(le)
    back, rate = P.read_audio_file(str(tmp_path / "le.sf"))
    assert rate == 44100.0 and np.array_equal(back[:, 0], y[:, 0].astype(np.float32) / np.float32(32768.0))
    assert lib.sgz_write_float_audio(str(tmp_path / "o.x").encode(), b"raw", N.ptr(x), 50, 2, 86.0) == N.SGZ_E_ARG


def test_secs_to_frames_is_the_references_rounding():
    # (s * sampleRate + 0.5).toLong, Strugatzki.scala:143 -- truncation towards zero, NaN -> 0
    cases = [(0.0, 44100.0, 0), (1.0, 44100.0, 44100), (0.5, 44100.0, 22050), (30.0, 44100.0, 1323000),
             (1e-5, 44100.0, 0), (1.2e-5, 44100.0, 1), (0.1, 48000.0, 4800), (2.0000113, 44100.0, 88200),
             (-1.0, 44100.0, -44099), (float("nan"), 44100.0, 0)]
    for s, sr, want in cases:
        assert lib.sgz_secs_to_frames(s, sr) == want
        if s == s:
            assert want == int(s * sr + 0.5)


def _from_seconds(**kw):
    a = dict(dir=b"db", meta=b"db/in_feat.xml", sr=44100.0, in_start=1.0, in_stop=31.0, in_temp=0.5, has_out=0,
             out_start=0.0, out_stop=0.0, out_temp=0.5, dur_min=0.5, dur_max=2.0, boost=8.0, nm=10, npf=1, spacing=0.0,
             norm=1)
    a.update(kw)
    out = C.c_void_p()
    rc = lib.sgz_correlate_config_from_seconds(a["dir"], a["meta"], a["sr"], a["in_start"], a["in_stop"], a["in_temp"],
                                               a["has_out"], a["out_start"], a["out_stop"], a["out_temp"], a["dur_min"],
                                               a["dur_max"], a["boost"], a["nm"], a["npf"], a["spacing"], a["norm"],
                                               C.byref(out))
    if rc != 0:
        return rc, lib.sgz_host_last_error().decode()
    s = C.string_at(out).decode()
    lib.sgz_free_string(out)
    return 0, s


def test_correlate_config_from_the_command_lines_seconds():
    rc, xml = _from_seconds()
    assert rc == 0
    r = ET.fromstring(xml)
    assert r.tag == "correlate" and r.findtext("database") == "db" and r.findtext("input") == "db/in_feat.xml"
    assert (r.find("punchIn").findtext("start"), r.find("punchIn").findtext("stop")) == ("44100", "1367100")
    assert r.find("punchOut") is None
    assert (r.findtext("minPunch"), r.findtext("maxPunch"), r.findtext("minSpacing")) == ("22050", "88200", "0")
    assert (r.findtext("numMatches"), r.findtext("numPerFile"), r.findtext("normalize")) == ("10", "1", "true")
    cfg = P.FeatureCorrelation.Config.from_xml(xml)  # and it is a document the processors take
    assert cfg.punchIn.span == P.Span(44100, 1367100) and cfg.maxBoost == 8.0
    rc, xml = _from_seconds(has_out=1, out_start=40.0, out_stop=45.25, out_temp=0.25, spacing=0.75)
    r = ET.fromstring(xml)
    assert (r.find("punchOut").findtext("start"), r.find("punchOut").findtext("stop")) == ("1764000", "1995525")
    assert r.find("punchOut").findtext("weight") == "0.25" and r.findtext("minSpacing") == "33075"
    # the reference's require(...) checks, Strugatzki.scala:147, 157, 160, 162
    for kw, msg in ((dict(in_stop=1.0), "Punch in span is empty"), (dict(dur_min=0.0), "Minimum duration is zero"),
                    (dict(dur_max=0.1), "Maximum duration is smaller"),
                    (dict(has_out=1, out_start=3.0, out_stop=3.0), "Punch out span is empty")):
        rc, err = _from_seconds(**kw)
        assert rc == N.SGZ_E_ARG and msg in err
