"""XML forms of configs and results -- the scenario of the reference's own suite
(src/test/scala/de/sciss/strugatzki/StrugatzkiSuite.scala:12-92): build configs with
perturbed defaults, transform to XML and back, expect equality.  The "back" goes
through the library's C++ parser and printer as well as through the Python mirror.
Also covers SelfSimilarity.Config and Break, which the reference suite leaves out.
"""
import os
from dataclasses import replace

import numpy as np
import pytest

from strugatzki_amd import processors as P
from strugatzki_amd.processors import (FeatureCorrelation, FeatureExtraction, FeatureSegmentation, SelfSimilarity,
                                       Span)


def _both_ways(kind, obj, cls):
    xml = obj.to_xml()
    assert cls.from_xml(xml) == obj                                # Python parser
    assert cls.from_xml(P.roundtrip_native(kind, xml)) == obj      # C++ fromXML -> toXML -> Python parser
    again = P.roundtrip_native(kind, P.roundtrip_native(kind, xml))
    assert again == P.roundtrip_native(kind, xml)                  # the C++ form is a fixed point


def test_feature_extraction_config_round_trip():
    fe = FeatureExtraction.Config()
    fe.audioInput = os.path.abspath("testing.aif")
    fe.featureOutput = "relative.aif"
    fe.metaOutput = None
    fe.numCoeffs += 1
    fe.fftSize += 1
    fe.fftOverlap += 1
    _both_ways("feature", fe, FeatureExtraction.Config)
    fe2 = replace(fe, metaOutput=os.path.dirname(fe.audioInput))
    _both_ways("feature", fe2, FeatureExtraction.Config)


def test_feature_correlation_config_round_trip():
    fc = FeatureCorrelation.Config()
    fc.databaseFolder = os.path.realpath("db")
    fc.metaInput = "rarara.xml"
    old = fc.punchIn
    fc.punchIn = FeatureCorrelation.Punch(Span(old.span.start + 1, old.span.stop + 2),
                                          float(np.float32(old.temporalWeight) + np.float32(0.11)))
    fc.punchOut = FeatureCorrelation.Punch(Span(555, 666), float(np.float32(0.1234)))
    fc.minPunch += 1
    fc.maxPunch += 2
    fc.normalize = not fc.normalize
    fc.maxBoost += 1
    fc.numMatches += 1
    fc.numPerFile += 1
    fc.minSpacing += 1
    _both_ways("correlate", fc, FeatureCorrelation.Config)
    fc2 = replace(fc, punchOut=None, normalize=not fc.normalize)
    _both_ways("correlate", fc2, FeatureCorrelation.Config)


def test_match_round_trip():
    fm1 = FeatureCorrelation.Match(float(np.float32(0.23)), "gaga.aif", Span(33, 44), -6.0, -7.0)
    fm2 = FeatureCorrelation.Match(float(np.float32(0.46)), os.path.abspath("rara.wav"), Span(666, 777), 8.0, 9.0)
    _both_ways("match", fm1, FeatureCorrelation.Match)
    _both_ways("match", fm2, FeatureCorrelation.Match)


def test_feature_segmentation_config_round_trip():
    fs = FeatureSegmentation.Config()
    fs.databaseFolder = os.path.realpath("db")
    fs.metaInput = "rarara.xml"
    fs.span = Span(1, 2)
    fs.corrLen += 1
    fs.temporalWeight = float(np.float32(fs.temporalWeight) + np.float32(0.1))
    fs.normalize = not fs.normalize
    fs.numBreaks += 1
    fs.minSpacing += 1
    _both_ways("segmentation", fs, FeatureSegmentation.Config)
    fs2 = replace(fs, span=Span.all(), normalize=not fs.normalize)
    _both_ways("segmentation", fs2, FeatureSegmentation.Config)
    _both_ways("segmentation", replace(fs, span=Span(start=5)), FeatureSegmentation.Config)
    _both_ways("segmentation", replace(fs, span=Span(stop=9)), FeatureSegmentation.Config)


def test_self_similarity_config_and_break_round_trip():
    ss = SelfSimilarity.Config()
    ss.metaInput2 = "other_feat.xml"
    ss.span = Span(100, 20000)
    ss.corrLen += 3
    ss.decimation = 4
    ss.colors = SelfSimilarity.GRAY
    ss.colorWarp = float(np.float32(0.7))
    ss.colorCeil = float(np.float32(0.95))
    ss.colorInv = True
    ss.normalize = False
    _both_ways("selfsimilarity", ss, SelfSimilarity.Config)
    _both_ways("selfsimilarity", replace(ss, metaInput2=None, span=Span.all(), colors="psycho"), SelfSimilarity.Config)
    _both_ways("break", FeatureSegmentation.Break(float(np.float32(0.125)), 123456789012), FeatureSegmentation.Break)


@pytest.mark.parametrize("kind,cls", [("feature", FeatureExtraction.Config), ("correlate", FeatureCorrelation.Config),
                                      ("segmentation", FeatureSegmentation.Config),
                                      ("selfsimilarity", SelfSimilarity.Config)])
def test_defaults_match_the_reference(kind, cls):
    """ConfigBuilder defaults: FeatureExtraction.scala:112-143, FeatureCorrelation.scala:168-206,
    FeatureSegmentation.scala:134-159, SelfSimilarity.scala:153-187."""
    native = cls.from_xml(P.default_config_xml(kind))
    assert native == cls()
    d = cls()
    if kind == "correlate":
        assert (d.databaseFolder, d.metaInput, d.punchIn, d.punchOut, d.minPunch, d.maxPunch, d.normalize, d.maxBoost,
                d.numMatches, d.numPerFile, d.minSpacing) == \
               ("database", "input_feat.xml", FeatureCorrelation.Punch(Span(0, 44100), 0.5), None, 22050, 88200, True,
                8.0, 1, 1, 0)
    if kind == "segmentation":
        assert (d.span, d.corrLen, d.temporalWeight, d.normalize, d.numBreaks, d.minSpacing) == \
               (Span.all(), 22050, 0.5, True, 1, 22050)
    if kind == "selfsimilarity":
        assert (d.imageOutput, d.corrLen, d.decimation, d.colors, d.colorWarp, d.colorCeil, d.colorInv) == \
               ("output_selfsim.png", 44100, 1, "psycho", 1.0, 1.0, False)
    if kind == "feature":
        assert (d.audioInput, d.numCoeffs, d.fftSize, d.fftOverlap, d.channelsBehavior) == ("input.aif", 13, 1024, 2, 0)


def test_xml_escaping_and_bad_documents():
    fc = FeatureCorrelation.Config(databaseFolder="a&b <c>", metaInput='q"uote.xml')
    _both_ways("correlate", fc, FeatureCorrelation.Config)
    with pytest.raises(P.N.NativeError):
        P.roundtrip_native("correlate", "<correlate><database>x</database></correlate>")  # missing numbers
    with pytest.raises(P.N.NativeError):
        P.roundtrip_native("selfsimilarity", SelfSimilarity.Config(colors="rainbow").to_xml())
    with pytest.raises(P.N.NativeError):
        P.roundtrip_native("nonsense", "<x/>")


def test_float_aiff_round_trip(tmp_path):
    rng = np.random.default_rng(1)
    frames = rng.standard_normal((37, 14)).astype(np.float32)
    frames[3, 2] = np.float32(np.nan)
    path = str(tmp_path / "x_feat.aif")
    P.write_float_aiff(path, frames, 44100.0 / 512)
    back, sr = P.read_audio_file(path)
    assert sr == 44100.0 / 512 and back.shape == frames.shape
    assert np.array_equal(back.view(np.uint32), frames.view(np.uint32))
    raw = open(path, "rb").read()
    assert raw[:4] == b"FORM" and raw[8:12] == b"AIFC" and b"fl32" in raw[:80]  # big-endian float AIFF-C
    assert int.from_bytes(raw[4:8], "big") == len(raw) - 8


def test_read_pcm_aiff_and_wave(tmp_path):
    import struct
    import wave
    pcm = (np.arange(-8, 8, dtype=np.int16) * 1000).reshape(8, 2)
    wpath = str(tmp_path / "a.wav")
    with wave.open(wpath, "wb") as w:
        w.setnchannels(2)
        w.setsampwidth(2)
        w.setframerate(48000)
        w.writeframes(pcm.tobytes())
    got, sr = P.read_audio_file(wpath)
    assert sr == 48000 and np.array_equal(got, pcm.astype(np.float32) / 32768)
    # plain 16-bit AIFF written by hand
    body = pcm.astype(">i2").tobytes()
    ext80 = bytes([0x40, 0x0E, 0xAC, 0x44, 0, 0, 0, 0, 0, 0])  # 44100.0
    comm = struct.pack(">hIh", 2, 8, 16) + ext80
    ssnd = struct.pack(">II", 0, 0) + body
    form = b"AIFF" + b"COMM" + struct.pack(">I", len(comm)) + comm + b"SSND" + struct.pack(">I", len(ssnd)) + ssnd
    apath = str(tmp_path / "a.aif")
    open(apath, "wb").write(b"FORM" + struct.pack(">I", len(form)) + form)
    got, sr = P.read_audio_file(apath)
    assert sr == 44100 and np.array_equal(got, pcm.astype(np.float32) / 32768)
    with pytest.raises(P.N.NativeError):
        P.read_audio_file(str(tmp_path / "missing.aif"))


def test_crosssimilarity_config_roundtrip():
    """<crosssimilarity> (CrossSimilarity.scala:182-246): Python mirror and C++ parser / printer agree; open
    spans are left out of the document."""
    from strugatzki_amd.processors import CrossSimilarity, Span, roundtrip_native
    cfgs = [CrossSimilarity.Config(),
            CrossSimilarity.Config("db & co", "a_feat.xml", "b<1>_feat.xml", "out.aif", "aiff", Span(100, 90000),
                                   Span(None, 44100), 0.25, False, 3.5),
            CrossSimilarity.Config(span1=Span(5, None))]
    for c in cfgs:
        xml = c.to_xml()
        assert CrossSimilarity.Config.from_xml(xml) == c
        back = roundtrip_native("crosssimilarity", xml)
        assert CrossSimilarity.Config.from_xml(back) == c
        assert ("<span1>" in back) == (c.span1 != Span()) and ("<span2>" in back) == (c.span2 != Span())
