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
    (25000, 1, 0, 2048, 4, 20), (9000, 1, 0, 256, 1, 5), (700, 1, 0, 1024, 2, 13), (1025, 1, 0, 1024, 2, 13)])
def test_gpu_features_match_the_oracle(ctx, n, channels, behavior, fft, overlap, coeffs):
    x = _signal(n, seed=n, channels=channels)
    want = O.extract(x, SR, fft, overlap, coeffs, behavior)
    got = ctx.extract(x, SR, fft, overlap, coeffs, behavior)
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=0, atol=TOL)
    if got.shape[0] > 1:
        # loudness spans a wider range than 0..1: relative check as well
        np.testing.assert_allclose(got[:, 0], want[:, 0], rtol=5e-4, atol=1e-5)


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
