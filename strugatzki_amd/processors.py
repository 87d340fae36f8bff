"""The reference's processor API, mirrored for Python callers.

`FeatureCorrelation`, `FeatureSegmentation`, `SelfSimilarity` and
`FeatureExtraction` carry a `Config` with the reference's field names, defaults
and XML element names (FeatureCorrelation.scala:168-273,
FeatureSegmentation.scala:134-212, SelfSimilarity.scala:153-283,
FeatureExtraction.scala:112-207) and a `run(ctx, config, observer)` that hands
the XML to the library's host side (sgz_run_*_xml), where the whole processor
body runs: files, norms, GPU kernels, selection.
"""
from __future__ import annotations

import ctypes as C
import os
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from typing import Callable, Optional
from xml.sax.saxutils import escape

import numpy as np

from . import _native as N
from .engine import Context

lib = N.lib
Observer = Optional[Callable[[float], bool]]


def _f(v: float) -> str:
    """float32 -> text that parses back to the same float32."""
    return repr(float(np.float32(v)))


def _b(v: bool) -> str:
    return "true" if v else "false"


def _take_string(p: C.c_void_p) -> str:
    try:
        return C.cast(p, C.c_char_p).value.decode()
    finally:
        lib.sgz_free_string(p)


def _host_check(rc: int) -> None:
    if rc != N.SGZ_OK:
        raise N.NativeError(rc, lib.sgz_host_last_error().decode())


def roundtrip_native(kind: str, xml: str) -> str:
    """Parses and re-prints a document with the library's C++ XML code (X.fromXML(x.toXML))."""
    out = C.c_void_p()
    _host_check(lib.sgz_config_roundtrip_xml(kind.encode(), xml.encode(), C.byref(out)))
    return _take_string(out)


def default_config_xml(kind: str) -> str:
    out = C.c_void_p()
    _host_check(lib.sgz_default_config_xml(kind.encode(), C.byref(out)))
    return _take_string(out)


def _callback(observer: Observer):
    if observer is None:
        return None
    return N.PROGRESS_FN(lambda p, _u: 1 if observer(p) is not False else 0)


@dataclass(frozen=True)
class Span:
    """de.sciss.span.Span.NonVoid: Span(a, b), Span.from(a), Span.until(b), Span.all."""
    start: Optional[int] = None
    stop: Optional[int] = None

    @staticmethod
    def all() -> "Span":
        return Span()

    def _children(self) -> str:
        return ("" if self.start is None else f"<start>{self.start}</start>") + \
               ("" if self.stop is None else f"<stop>{self.stop}</stop>")

    @staticmethod
    def _from(node) -> "Span":
        if node is None:
            return Span()
        a, b = node.find("start"), node.find("stop")
        return Span(None if a is None else int(a.text), None if b is None else int(b.text))


def _text(node, name: str) -> str:
    c = node.find(name)
    return "" if c is None or c.text is None else c.text


class FeatureExtraction:
    MIX, FIRST, LAST = 0, 1, 2  # ChannelsBehavior ids, FeatureExtraction.scala:36-52

    @dataclass
    class Config:
        audioInput: str = "input.aif"
        featureOutput: str = ""
        metaOutput: Optional[str] = None
        numCoeffs: int = 13
        fftSize: int = 1024
        fftOverlap: int = 2
        channelsBehavior: int = 0

        def to_xml(self) -> str:
            return (f"<feature><input>{escape(self.audioInput)}</input><output>{escape(self.featureOutput)}</output>"
                    f"<meta>{escape(self.metaOutput or '')}</meta><numCoeffs>{self.numCoeffs}</numCoeffs>"
                    f"<fftSize>{self.fftSize}</fftSize><fftOverlap>{self.fftOverlap}</fftOverlap>"
                    f"<channels>{self.channelsBehavior}</channels></feature>")

        @staticmethod
        def from_xml(xml: str) -> "FeatureExtraction.Config":
            r = ET.fromstring(xml)
            meta = _text(r, "meta")
            ch = _text(r, "channels")
            return FeatureExtraction.Config(_text(r, "input"), _text(r, "output"), meta or None,
                                            int(_text(r, "numCoeffs")), int(_text(r, "fftSize")),
                                            int(_text(r, "fftOverlap")), int(ch) if ch else 0)

    @staticmethod
    def run(ctx: Context, config: "FeatureExtraction.Config", observer: Observer = None) -> None:
        cb = _callback(observer)
        _host_check(lib.sgz_run_extraction_xml(ctx._h, config.to_xml().encode(), cb, None))


class FeatureStats:
    """FeatureStats.scala:22-34: Config = the feature files, Product = per channel (min, max)."""

    NORMALIZE_NAME = "feat_norms.aif"  # Strugatzki.NormalizeName

    @staticmethod
    def run(ctx: Context, config: list[str], observer: Observer = None) -> list[tuple[float, float]]:
        """Per channel the smallest 1st and the largest 99th percentile over the given feature files
        (impl/FeatureStatsImpl.scala:31-54).  Replaces the files of the context's database."""
        paths = [os.fspath(p).encode() for p in config]
        arr = (C.c_char_p * max(len(paths), 1))(*paths)
        out = np.zeros((N.SGZ_MAX_CHANNELS, 2), dtype=np.float64)
        nch = C.c_int(0)
        cb = _callback(observer)
        _host_check(lib.sgz_run_feature_stats(ctx._h, arr, len(paths), cb, None, N.ptr(out), N.SGZ_MAX_CHANNELS,
                                              C.byref(nch)))
        return [(float(a), float(b)) for a, b in out[: nch.value]]

    @staticmethod
    def write_norms(path: str, spans: list[tuple[float, float]]) -> None:
        """The normalisation file the searches read (Strugatzki.scala:414-429)."""
        a = np.ascontiguousarray(spans, dtype=np.float64)
        _host_check(lib.sgz_write_norms_file(os.fspath(path).encode(), N.ptr(a), a.shape[0]))


class FeatureCorrelation:
    @dataclass(frozen=True)
    class Punch:
        span: Span = Span(0, 44100)
        temporalWeight: float = 0.5

        def _children(self) -> str:
            return f"<start>{self.span.start}</start><stop>{self.span.stop}</stop><weight>{_f(self.temporalWeight)}</weight>"

        @staticmethod
        def _from(node) -> "FeatureCorrelation.Punch":
            return FeatureCorrelation.Punch(Span(int(_text(node, "start")), int(_text(node, "stop"))),
                                            float(np.float32(_text(node, "weight"))))

    @dataclass(frozen=True)
    class Match:
        sim: float
        file: str
        punch: Span
        boostIn: float
        boostOut: float

        def to_xml(self) -> str:
            return (f"<match><sim>{_f(self.sim)}</sim><file>{escape(self.file)}</file><start>{self.punch.start}</start>"
                    f"<stop>{self.punch.stop}</stop><boostIn>{_f(self.boostIn)}</boostIn>"
                    f"<boostOut>{_f(self.boostOut)}</boostOut></match>")

        @staticmethod
        def from_xml(xml) -> "FeatureCorrelation.Match":
            r = ET.fromstring(xml) if isinstance(xml, str) else xml
            return FeatureCorrelation.Match(float(np.float32(_text(r, "sim"))), _text(r, "file"),
                                            Span(int(_text(r, "start")), int(_text(r, "stop"))),
                                            float(np.float32(_text(r, "boostIn"))), float(np.float32(_text(r, "boostOut"))))

    @dataclass
    class Config:
        databaseFolder: str = "database"
        metaInput: str = "input_feat.xml"
        punchIn: "FeatureCorrelation.Punch" = field(default_factory=lambda: FeatureCorrelation.Punch())
        punchOut: Optional["FeatureCorrelation.Punch"] = None
        minPunch: int = 22050
        maxPunch: int = 88200
        normalize: bool = True
        maxBoost: float = 8.0
        numMatches: int = 1
        numPerFile: int = 1
        minSpacing: int = 0

        def to_xml(self) -> str:
            po = "" if self.punchOut is None else f"<punchOut>{self.punchOut._children()}</punchOut>"
            return (f"<correlate><database>{escape(self.databaseFolder)}</database><input>{escape(self.metaInput)}</input>"
                    f"<punchIn>{self.punchIn._children()}</punchIn>{po}<minPunch>{self.minPunch}</minPunch>"
                    f"<maxPunch>{self.maxPunch}</maxPunch><normalize>{_b(self.normalize)}</normalize>"
                    f"<maxBoost>{_f(self.maxBoost)}</maxBoost><numMatches>{self.numMatches}</numMatches>"
                    f"<numPerFile>{self.numPerFile}</numPerFile><minSpacing>{self.minSpacing}</minSpacing></correlate>")

        @staticmethod
        def from_xml(xml: str) -> "FeatureCorrelation.Config":
            r = ET.fromstring(xml)
            po = r.find("punchOut")
            return FeatureCorrelation.Config(
                _text(r, "database"), _text(r, "input"), FeatureCorrelation.Punch._from(r.find("punchIn")),
                None if po is None else FeatureCorrelation.Punch._from(po), int(_text(r, "minPunch")),
                int(_text(r, "maxPunch")), _text(r, "normalize").strip() == "true",
                float(np.float32(_text(r, "maxBoost"))), int(_text(r, "numMatches")), int(_text(r, "numPerFile")),
                int(_text(r, "minSpacing")))

    @staticmethod
    def run(ctx: Context, config: "FeatureCorrelation.Config", observer: Observer = None
            ) -> list["FeatureCorrelation.Match"]:
        """Matches sorted by descending similarity (FeatureCorrelation.scala:30-33).  `ctx` may be a list of
        contexts (one per GPU): the database is then dealt over them (sgz_run_correlation_xml_multi)."""
        out = C.c_void_p()
        cb = _callback(observer)
        if isinstance(ctx, (list, tuple)):
            handles = (C.c_void_p * len(ctx))(*[c._h for c in ctx])
            _host_check(lib.sgz_run_correlation_xml_multi(handles, len(ctx), config.to_xml().encode(), cb, None,
                                                          C.byref(out)))
        else:
            _host_check(lib.sgz_run_correlation_xml(ctx._h, config.to_xml().encode(), cb, None, C.byref(out)))
        root = ET.fromstring(_take_string(out))
        return [FeatureCorrelation.Match.from_xml(m) for m in root.findall("match")]


class FeatureSegmentation:
    @dataclass(frozen=True)
    class Break:
        sim: float
        pos: int

        def to_xml(self) -> str:
            return f"<break><sim>{_f(self.sim)}</sim><pos>{self.pos}</pos></break>"

        @staticmethod
        def from_xml(xml) -> "FeatureSegmentation.Break":
            r = ET.fromstring(xml) if isinstance(xml, str) else xml
            return FeatureSegmentation.Break(float(np.float32(_text(r, "sim"))), int(_text(r, "pos")))

    @dataclass
    class Config:
        databaseFolder: str = "database"
        metaInput: str = "input_feat.xml"
        span: Span = Span()
        corrLen: int = 22050
        temporalWeight: float = 0.5
        normalize: bool = True
        numBreaks: int = 1
        minSpacing: int = 22050

        def to_xml(self) -> str:
            return (f"<segmentation><database>{escape(self.databaseFolder)}</database><input>{escape(self.metaInput)}"
                    f"</input><span>{self.span._children()}</span><corr>{self.corrLen}</corr>"
                    f"<weight>{_f(self.temporalWeight)}</weight><normalize>{_b(self.normalize)}</normalize>"
                    f"<numBreaks>{self.numBreaks}</numBreaks><minSpacing>{self.minSpacing}</minSpacing></segmentation>")

        @staticmethod
        def from_xml(xml: str) -> "FeatureSegmentation.Config":
            r = ET.fromstring(xml)
            return FeatureSegmentation.Config(_text(r, "database"), _text(r, "input"), Span._from(r.find("span")),
                                              int(_text(r, "corr")), float(np.float32(_text(r, "weight"))),
                                              _text(r, "normalize").strip() == "true", int(_text(r, "numBreaks")),
                                              int(_text(r, "minSpacing")))

    @staticmethod
    def run(ctx: Context, config: "FeatureSegmentation.Config", observer: Observer = None
            ) -> list["FeatureSegmentation.Break"]:
        """Breaks sorted by ascending similarity (FeatureSegmentation.scala:32-35)."""
        out = C.c_void_p()
        cb = _callback(observer)
        _host_check(lib.sgz_run_segmentation_xml(ctx._h, config.to_xml().encode(), cb, None, C.byref(out)))
        root = ET.fromstring(_take_string(out))
        return [FeatureSegmentation.Break.from_xml(b) for b in root.findall("break")]


class SelfSimilarity:
    GRAY, PSYCHO = "gray", "psycho"  # ColorScheme logical names, SelfSimilarity.scala:30-56

    @dataclass
    class Config:
        databaseFolder: str = "database"
        metaInput: str = "input_feat.xml"
        metaInput2: Optional[str] = None
        imageOutput: str = "output_selfsim.png"
        span: Span = Span()
        corrLen: int = 44100
        decimation: int = 1
        temporalWeight: float = 0.5
        colors: str = "psycho"
        colorWarp: float = 1.0
        colorCeil: float = 1.0
        colorInv: bool = False
        normalize: bool = True

        def to_xml(self) -> str:
            i2 = "" if self.metaInput2 is None else f"<input2>{escape(self.metaInput2)}</input2>"
            sp = "" if self.span == Span() else f"<span>{self.span._children()}</span>"
            return (f"<selfsimilarity><database>{escape(self.databaseFolder)}</database><input>{escape(self.metaInput)}"
                    f"</input>{i2}<output>{escape(self.imageOutput)}</output>{sp}<corr>{self.corrLen}</corr>"
                    f"<decimation>{self.decimation}</decimation><weight>{_f(self.temporalWeight)}</weight>"
                    f"<colors>{self.colors}</colors><colorWarp>{_f(self.colorWarp)}</colorWarp>"
                    f"<colorCeil>{_f(self.colorCeil)}</colorCeil><colorInv>{_b(self.colorInv)}</colorInv>"
                    f"<normalize>{_b(self.normalize)}</normalize></selfsimilarity>")

        @staticmethod
        def from_xml(xml: str) -> "SelfSimilarity.Config":
            r = ET.fromstring(xml)
            i2 = r.find("input2")
            return SelfSimilarity.Config(
                _text(r, "database"), _text(r, "input"), None if i2 is None else (i2.text or ""), _text(r, "output"),
                Span._from(r.find("span")), int(_text(r, "corr")), int(_text(r, "decimation")),
                float(np.float32(_text(r, "weight"))), _text(r, "colors").strip(),
                float(np.float32(_text(r, "colorWarp"))), float(np.float32(_text(r, "colorCeil"))),
                _text(r, "colorInv").strip() == "true", _text(r, "normalize").strip() == "true")

    @staticmethod
    def run(ctx: Context, config: "SelfSimilarity.Config", observer: Observer = None) -> None:
        cb = _callback(observer)
        _host_check(lib.sgz_run_selfsimilarity_xml(ctx._h, config.to_xml().encode(), cb, None))

    @staticmethod
    def rows(ctx: Context, config: "SelfSimilarity.Config", x_begin: int = 0, x_end: int | None = None) -> np.ndarray:
        """Rows [x_begin, x_end) (default: all) of the similarity matrix behind the raster, no colour scheme applied
        (sgz_run_selfsimilarity_rows_xml): what a caller with a palette of its own colours, band by band."""
        import ctypes as C
        xml = config.to_xml().encode()
        ext = C.c_int(0)
        _host_check(lib.sgz_run_selfsimilarity_rows_xml(ctx._h, xml, 0, 0, None, C.byref(ext)))
        if x_end is None:
            x_end = ext.value
        out = np.zeros((max(x_end - x_begin, 0), ext.value), dtype=np.float32)
        if out.shape[0] or x_end != x_begin:
            _host_check(lib.sgz_run_selfsimilarity_rows_xml(ctx._h, xml, x_begin, x_end, N.ptr(out) if out.size else None,
                                                            C.byref(ext)))
        return out


class CrossSimilarity:
    @dataclass
    class Config:
        """CrossSimilarity.scala:30-246; XML document <crosssimilarity>."""
        databaseFolder: str = "database"
        metaInput1: str = "input1_feat.xml"
        metaInput2: str = "input2_feat.xml"
        audioOutput: str = "output.aif"
        audioOutputType: str = "aiff"
        span1: Span = Span()
        span2: Span = Span()
        temporalWeight: float = 0.5
        normalize: bool = True
        maxBoost: float = 8.0

        def to_xml(self) -> str:
            def sp(tag, s):
                if s.start is None and s.stop is None:
                    return ""
                a = "" if s.start is None else f"<start>{s.start}</start>"
                b = "" if s.stop is None else f"<stop>{s.stop}</stop>"
                return f"<{tag}>{a}{b}</{tag}>"
            return (f"<crosssimilarity><database>{escape(self.databaseFolder)}</database>"
                    f"<input1>{escape(self.metaInput1)}</input1><input2>{escape(self.metaInput2)}</input2>"
                    f"<output>{escape(self.audioOutput)}</output><outputType>{escape(self.audioOutputType)}</outputType>"
                    f"{sp('span1', self.span1)}{sp('span2', self.span2)}<weight>{_f(self.temporalWeight)}</weight>"
                    f"<normalize>{_b(self.normalize)}</normalize><maxBoost>{_f(self.maxBoost)}</maxBoost>"
                    f"</crosssimilarity>")

        @staticmethod
        def from_xml(xml: str) -> "CrossSimilarity.Config":
            r = ET.fromstring(xml)

            def sp(tag):
                e = r.find(tag)
                if e is None:
                    return Span()
                a, b = e.find("start"), e.find("stop")
                return Span(None if a is None else int(a.text), None if b is None else int(b.text))
            return CrossSimilarity.Config(_text(r, "database"), _text(r, "input1"), _text(r, "input2"),
                                          _text(r, "output"), _text(r, "outputType"), sp("span1"), sp("span2"),
                                          float(np.float32(_text(r, "weight"))),
                                          _text(r, "normalize").strip() == "true",
                                          float(np.float32(_text(r, "maxBoost"))))

    @staticmethod
    def run(ctx: Context, config: "CrossSimilarity.Config", observer: Observer = None) -> None:
        """Writes the similarity curve to config.audioOutput.  Replaces the files of the context's database."""
        cb = _callback(observer)
        _host_check(lib.sgz_run_crosssimilarity_xml(ctx._h, config.to_xml().encode(), cb, None))


# ---- feature / audio files ------------------------------------------------------
def write_float_aiff(path: str, frames: np.ndarray, sample_rate: float) -> None:
    a = np.ascontiguousarray(frames, dtype=np.float32)
    if a.ndim == 1:
        a = a[:, None]
    _host_check(lib.sgz_write_float_aiff(path.encode(), N.ptr(a), a.shape[0], a.shape[1], sample_rate))


def read_audio_file(path: str) -> tuple[np.ndarray, float]:
    n, ch, sr = C.c_int64(), C.c_int(), C.c_double()
    _host_check(lib.sgz_read_audio_file(path.encode(), None, 0, C.byref(n), C.byref(ch), C.byref(sr)))
    out = np.empty((n.value, ch.value), dtype=np.float32)
    _host_check(lib.sgz_read_audio_file(path.encode(), N.ptr(out), n.value, None, None, None))
    return out, sr.value
