"""Feature extraction: framing follows the reference, the spectral model is this build's own
(scsynth's MFCC / Loudness UGens are not part of the reference sources: parity unpinned).
The GPU kernels (fp32) are checked against the fp64 statement of the same model in the oracle."""
import os

import numpy as np
import pytest

from common import O

SR = 44100.0
TOL = 2e-4  # absolute, on features of order 0..1 (fp32 FFT of 1024 points vs fp64)


def _signal(n, seed=0, channels=1):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / SR
    x = 0.4 * np.sin(2 * np.pi * 440 * t) * (0.5 + 0.5 * np.sin(2 * np.pi * 0.7 * t)) + 0.1 * rng.standard_normal(n)
    if channels == 1:
        return x.astype(np.float32)[:, None]
    y = 0.3 * np.sin(2 * np.pi * 1234 * t) + 0.05 * rng.standard_normal(n)
    return np.stack([x, y], axis=1).astype(np.float32)


# ------------------------------------------------------------------ oracle (CPU)
@pytest.mark.parametrize("n,step,want", [(0, 512, 0), (1, 512, 0), (512, 512, 0), (513, 512, 1), (1024, 512, 1),
                                         (158760000, 512, 310078), (13230000, 512, 25839)])
def test_frame_count_follows_the_nrt_render(n, step, want):
    """ceil(S / step) - 1 (impl/NonRealtimeProcessor.scala:93,107-108)."""
    assert O.lib().sgzo_extract_num_frames(n, step) == want


def test_oracle_layout_and_first_frame():
    x = _signal(8000)
    f = O.extract(x, SR)
    assert f.shape == (O.lib().sgzo_extract_num_frames(8000, 512), 14)
    assert np.all(f[0] == 0)              # stream index 1: no FFT frame complete yet
    assert np.all(f[1:, 0] > 0)           # loudness / 32
    assert np.all((f[1:, 1:] > 0) & (f[1:, 1:] < 1))


def test_oracle_loudness_grows_with_level_and_mix_sums_channels():
    x = _signal(6000, channels=2)
    quiet = O.extract(x * np.float32(0.1), SR)
    loud = O.extract(x, SR)
    assert np.all(loud[2:, 0] > quiet[2:, 0])
    mixed = O.extract(x, SR, channels=0)
    summed = O.extract(x.sum(axis=1, keepdims=True), SR)
    np.testing.assert_allclose(mixed, summed, atol=1e-6)
    first = O.extract(x, SR, channels=1)
    last = O.extract(x, SR, channels=2)
    np.testing.assert_allclose(first, O.extract(x[:, :1], SR), atol=0)
    np.testing.assert_allclose(last, O.extract(x[:, 1:], SR), atol=0)


def test_oracle_sine_peaks_in_the_right_band():
    """A 1 kHz sine must put its energy into the mel band around 1 kHz: checks band edges + FFT."""
    n = 8192
    x = (0.5 * np.sin(2 * np.pi * 1000 * np.arange(n) / SR)).astype(np.float32)[:, None]
    edges = (np.zeros(44, dtype=np.float64))
    import ctypes as C
    O.lib().sgzo_mel_edges.argtypes = [C.c_double, C.c_int, C.POINTER(C.c_double)]
    O.lib().sgzo_mel_edges(SR, 1024, edges.ctypes.data_as(C.POINTER(C.c_double)))
    hz = edges * SR / 1024
    assert hz[0] == 0 and abs(hz[-1] - SR / 2) < 1e-6 and np.all(np.diff(hz) > 0)
    f42 = O.extract(x, SR, num_coeffs=42)
    # invert the DCT-II of a steady frame to recover the band dB values
    c = (f42[5, 1:] - 0.5) * 2 * 100 * 42
    j = np.arange(42)
    basis = np.cos(np.pi / 42 * (j[None, :] + 0.5) * j[:, None])
    db = np.linalg.solve(basis, c)
    k = int(np.argmax(db))
    assert hz[k] < 1000 < hz[k + 2]


# ------------------------------------------------------------------ GPU vs oracle
@pytest.mark.gpu
@pytest.mark.parametrize("n,channels,behavior,fft,overlap,coeffs", [
    (30000, 1, 0, 1024, 2, 13), (30000, 2, 0, 1024, 2, 13), (20000, 2, 1, 1024, 2, 13), (20000, 2, 2, 1024, 2, 13),
    (25000, 1, 0, 2048, 4, 20), (9000, 1, 0, 256, 1, 5), (700, 1, 0, 1024, 2, 13), (1025, 1, 0, 1024, 2, 13),
    (2500, 1, 0, 1024, 1024, 13), (2500, 2, 0, 1024, 1024, 13), (9000, 2, 0, 1024, 8, 13), (9000, 3, 0, 1024, 4, 13)])
def test_gpu_features_match_the_oracle(ctx, n, channels, behavior, fft, overlap, coeffs):
    x = _signal(n, seed=n, channels=channels)
    want = O.extract(x, SR, fft, overlap, coeffs, behavior)
    got = ctx.extract(x, SR, fft, overlap, coeffs, behavior)
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=0, atol=TOL)
    if got.shape[0] > 1:
        # loudness spans a wider range than 0..1: relative check as well
        np.testing.assert_allclose(got[:, 0], want[:, 0], rtol=5e-4, atol=1e-5)


def _nan_signal(n, fft):
    """Samples that are not numbers, placed so that the spoilt feature frames lie at the start of the file, across
    the first two chunk boundaries of the feature writer (output frames 1023 and 2047) and in the middle of a chunk."""
    x = _signal(n, seed=3)
    step = fft // 2
    for frame in (0, 1, 400, 401, 402, 1021, 1022, 1023, 1024, 2046, 2047, 2048, 2500):
        if (frame + 1) * step < n:
            x[frame * step + 5, 0] = np.nan
    return x


def test_oracle_holds_the_last_value_over_nans_and_starts_each_chunk_at_zero():
    """fixNaNs of the feature writer (impl/NonRealtimeProcessor.scala:176-190): per chunk of 1024 stream frames
    (the first without stream frame 0) and channel, a NaN becomes the last value that was none, 0 at the start of
    a chunk."""
    n, fft = 1_400_000, 1024
    x = _nan_signal(n, fft)
    clean = O.extract(_signal(n, seed=3), SR, fft, 2, 13, 0)
    got = O.extract(x, SR, fft, 2, 13, 0)
    assert not np.isnan(got).any()
    # a spoilt sample spoils the two FFT frames that contain it: output frames f and f + 1 of sample f * step + 5
    assert np.array_equal(got[400:404], np.repeat(got[399:400], 4, axis=0))       # 400 .. 403 hold frame 399
    assert np.array_equal(got[404], clean[404])
    assert np.array_equal(got[0:3], np.zeros((3, 14), np.float32))                # nothing to hold yet
    assert np.array_equal(got[1021:1023], np.repeat(got[1020:1021], 2, axis=0))   # end of chunk 0
    assert np.array_equal(got[1023:1026], np.zeros((3, 14), np.float32))          # chunk 1 starts afresh
    assert np.array_equal(got[2046], got[2045])
    assert np.array_equal(got[2047:2050], np.zeros((3, 14), np.float32))          # chunk 2 as well
    assert np.array_equal(got[2500:2502], np.repeat(got[2499:2500], 2, axis=0))


@pytest.mark.gpu
@pytest.mark.parametrize("fft,coeffs", [(1024, 13), (512, 7)])
def test_gpu_holds_the_last_value_over_nans(ctx, fft, coeffs):
    n = 1_400_000
    x = _nan_signal(n, fft)
    want = O.extract(x, SR, fft, 2, coeffs, 0)
    got = ctx.extract(x, SR, fft, 2, coeffs, 0)
    assert not np.isnan(got).any()
    np.testing.assert_allclose(got, want, rtol=0, atol=TOL)
    # the frames that were replaced are exact copies of the frame before (or zeros), not approximations
    for f in (400, 1023, 2047, 2500):
        assert np.array_equal(got[f], got[f - 1]) or not got[f].any()


@pytest.mark.gpu
def test_gpu_silence_and_empty_input(ctx):
    got = ctx.extract(np.zeros((5000, 1), np.float32), SR)
    want = O.extract(np.zeros((5000, 1), np.float32), SR)
    np.testing.assert_allclose(got, want, atol=1e-6)
    assert ctx.extract(np.zeros((100, 1), np.float32), SR).shape == (0, 14)
    import strugatzki_amd as S
    with pytest.raises(S.NativeError):
        ctx.extract(np.zeros((5000, 1), np.float32), SR, fft_size=1000)


@pytest.mark.gpu
def test_feature_extraction_processor_writes_feature_file_and_meta(ctx, tmp_path):
    import wave
    from strugatzki_amd import processors as P
    from strugatzki_amd.processors import FeatureExtraction
    x = _signal(40000, seed=4, channels=2)
    wav = str(tmp_path / "in.wav")
    pcm16 = np.clip(np.round(x * 32767), -32768, 32767).astype(np.int16)
    with wave.open(wav, "wb") as w:
        w.setnchannels(2)
        w.setsampwidth(2)
        w.setframerate(44100)
        w.writeframes(pcm16.tobytes())
    feat, meta = str(tmp_path / "in_feat.aif"), str(tmp_path / "in_feat.xml")
    cfg = FeatureExtraction.Config(wav, feat, meta)
    FeatureExtraction.run(ctx, cfg)
    got, rate = P.read_audio_file(feat)
    assert rate == 44100.0 / 512 and got.shape == (40000 // 512 + 1 - 1, 14)
    want = O.extract(pcm16.astype(np.float32) / 32768, SR)
    np.testing.assert_allclose(got, want, rtol=0, atol=TOL)
    assert FeatureExtraction.Config.from_xml(open(meta).read()) == cfg
    # the written pair is a valid database entry for the other processors
    from strugatzki_amd.processors import FeatureSegmentation
    os.makedirs(tmp_path / "db", exist_ok=True)
    br = FeatureSegmentation.run(ctx, FeatureSegmentation.Config(str(tmp_path / "db"), meta, normalize=False,
                                                                 corrLen=4096, numBreaks=2))
    assert len(br) == 2


# ------------------------------------------------------------------ samples as audio files hold them
def _layouts(x):
    """float samples in [-1, 1) -> {SGZ_PCM_* code: (bytes, the float32 values those bytes stand for)}."""
    out = {}
    out[0] = (x.astype("<f4").tobytes(), x)
    out[1] = (x.astype(">f4").tobytes(), x)
    i16 = np.clip(np.round(x * 32768), -32768, 32767).astype(np.int16)
    out[2] = (i16.astype("<i2").tobytes(), i16.astype(np.float32) / np.float32(32768))
    out[3] = (i16.astype(">i2").tobytes(), out[2][1])
    i24 = np.clip(np.round(x.astype(np.float64) * 8388608), -8388608, 8388607).astype(np.int32)
    le = i24.astype("<i4").view(np.uint8).reshape(-1, 4)[:, :3]
    out[4] = (le.tobytes(), i24.astype(np.float32) / np.float32(8388608))
    out[5] = (le[:, ::-1].tobytes(), out[4][1])
    i32 = np.clip(np.round(x.astype(np.float64) * 2147483648), -2147483648, 2147483647).astype(np.int64).astype(np.int32)
    out[6] = (i32.astype("<i4").tobytes(), (i32.astype(np.float64) / 2147483648.0).astype(np.float32))
    out[7] = (i32.astype(">i4").tobytes(), out[6][1])
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("n,channels", [(30011, 1), (20005, 2), (8, 1), (0, 1)])
def test_gpu_sample_layouts_decode_on_the_device(ctx, n, channels):
    """16-, 24-, 32-bit integers and floats of either byte order, uploaded as they are: the features are those of
    the same values handed over as float32, bit for bit (an integer of b bits counts as v / 2^(b-1), the reading of
    the reference's audio file layer); lengths that are no multiple of the 8 samples a thread decodes."""
    x = _signal(max(n, 1), seed=n + 1, channels=channels)[:n]
    x[:4] = np.array([[-1.0, 0.999999, -0.5, 0.25][: channels]] * 4, np.float32)[: min(4, n)]
    for code, (raw, vals) in _layouts(x).items():
        want = ctx.extract(np.ascontiguousarray(vals, np.float32), SR)
        got = ctx.extract_pcm(raw, code, n, channels, SR)
        assert got.shape == want.shape, code
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), code
    import strugatzki_amd as S
    with pytest.raises(S.NativeError):
        ctx.extract_pcm(np.zeros(64, np.uint8), 8, 4, 1, SR)  # no such layout


@pytest.mark.gpu
@pytest.mark.parametrize("channels", [2, 3])
@pytest.mark.parametrize("behavior", [0, 1, 2])
def test_gpu_16_bit_samples_read_in_place_mix_like_floats(ctx, channels, behavior):
    """16-bit input of either byte order is read by the extraction kernel itself (no decoded copy): with every way of
    taking the channels (mix, first, last) the features are those of the same values as float32, bit for bit."""
    x = _signal(40000, seed=70 + channels, channels=2)
    if channels == 3:
        x = np.ascontiguousarray(np.concatenate([x, x[::-1, :1] * np.float32(0.5)], axis=1))
    raw, vals = _layouts(x)[2]
    want = ctx.extract(np.ascontiguousarray(vals, np.float32), SR, channels_behavior=behavior)
    for code in (2, 3):  # little- and big-endian
        got = ctx.extract_pcm(_layouts(x)[code][0], code, x.shape[0], channels, SR, channels_behavior=behavior)
        assert got.shape == want.shape and got.shape[0] > 70
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), code
    if channels == 2 and behavior != 2:
        # two channels have a load pattern of their own; a silent third channel sends the same samples down the
        # path for any number of channels
        padded = np.ascontiguousarray(np.concatenate([vals, np.zeros_like(vals[:, :1])], axis=1))
        assert np.array_equal(ctx.extract(padded, SR, channels_behavior=behavior).view(np.uint32), want.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("channels", [1, 2])
def test_gpu_16_bit_samples_with_a_hop_of_one_sample(ctx, channels):
    """fftOverlap = fftSize: windows start on odd samples, where no two 16-bit samples share an aligned word."""
    x = _signal(2600, seed=5, channels=channels)
    for code in (2, 3):
        raw, vals = _layouts(x)[code]
        want = ctx.extract(np.ascontiguousarray(vals, np.float32), SR, 1024, 1024)
        got = ctx.extract_pcm(raw, code, x.shape[0], channels, SR, 1024, 1024)
        assert got.shape == want.shape and got.shape[0] > 1500
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), code


@pytest.mark.gpu
def test_gpu_random_layouts_channel_counts_and_hops(ctx):
    """Random draws of length, channel count, way of taking the channels, hop and sample layout: the features from
    the file's bytes are those from the float values they stand for, bit for bit."""
    rng = np.random.default_rng(2024)
    for case in range(60):
        channels = int(rng.integers(1, 5))
        n = int(rng.integers(0, 30000))
        overlap = int(rng.choice([1, 2, 2, 2, 4, 8, 16, 64]))
        behavior = int(rng.integers(0, 3))
        code = int(rng.integers(0, 8))
        x = (0.3 * rng.standard_normal((max(n, 1), channels))).astype(np.float32).clip(-1, 0.99999)[:n]
        raw, vals = _layouts(x)[code]
        want = ctx.extract(np.ascontiguousarray(vals, np.float32).reshape(n, channels), SR, 1024, overlap, 13, behavior)
        got = ctx.extract_pcm(raw, code, n, channels, SR, 1024, overlap, 13, behavior)
        what = (case, n, channels, overlap, behavior, code)
        assert got.shape == want.shape, what
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), what


@pytest.mark.gpu
def test_gpu_integer_arrays_go_up_as_they_are(ctx):
    x = _signal(25000, seed=12, channels=2)
    i16 = np.round(x * 32767).astype(np.int16)
    want = ctx.extract(i16.astype(np.float32) / np.float32(32768), SR)
    for a in (i16, i16.astype(">i2"), i16[:, ::-1][:, ::-1]):
        got = ctx.extract(a, SR)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    i32 = (i16.astype(np.int32) << 16) + 12345
    want = ctx.extract((i32.astype(np.float64) / 2147483648.0).astype(np.float32), SR)
    assert np.array_equal(ctx.extract(i32, SR).view(np.uint32), want.view(np.uint32))
    np.testing.assert_allclose(want, O.extract((i32.astype(np.float64) / 2147483648.0).astype(np.float32), SR),
                               rtol=0, atol=TOL)


def _write_next(path, payload, encoding, channels):
    import struct
    with open(path, "wb") as f:
        f.write(b".snd" + struct.pack(">5I", 24, len(payload), encoding, 44100, channels) + payload)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["wav16", "wav24", "wav32", "aiff16", "aiff24", "aiff32", "next_f32", "next_f64",
                                  "wav8"])
def test_feature_extraction_reads_every_file_layout(ctx, tmp_path, kind):
    """FeatureExtraction over files of each sample layout: those the device decodes go up as the file's bytes, the
    others (8 bit, 64-bit floats) decoded by the reader -- the same features as from the decoded values."""
    import aifc
    import wave
    from strugatzki_amd import processors as P
    from strugatzki_amd.processors import FeatureExtraction
    x = _signal(30001, seed=21, channels=2)
    lay = _layouts(x)
    path = str(tmp_path / ("in." + kind))
    if kind.startswith("wav") and kind != "wav8":
        bits = int(kind[3:])
        raw, vals = lay[{16: 2, 24: 4, 32: 6}[bits]]
        with wave.open(path, "wb") as w:
            w.setnchannels(2); w.setsampwidth(bits // 8); w.setframerate(44100); w.writeframes(raw)
    elif kind == "wav8":
        u8 = np.clip(np.round(x * 128) + 128, 0, 255).astype(np.uint8)
        raw, vals = u8.tobytes(), (u8.astype(np.float32) - 128) / np.float32(128)
        with wave.open(path, "wb") as w:
            w.setnchannels(2); w.setsampwidth(1); w.setframerate(44100); w.writeframes(raw)
    elif kind.startswith("aiff"):
        bits = int(kind[4:])
        raw, vals = lay[{16: 3, 24: 5, 32: 7}[bits]]
        with aifc.open(path, "wb") as w:
            w.setnchannels(2); w.setsampwidth(bits // 8); w.setframerate(44100); w.writeframes(raw)
    elif kind == "next_f32":
        raw, vals = lay[1]
        _write_next(path, raw, 6, 2)
    else:
        raw, vals = x.astype(">f8").tobytes(), x
        _write_next(path, raw, 7, 2)
    feat = str(tmp_path / "out_feat.aif")
    FeatureExtraction.run(ctx, FeatureExtraction.Config(path, feat, None))
    got, _ = P.read_audio_file(feat)
    want = ctx.extract(np.ascontiguousarray(vals, np.float32), SR)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
