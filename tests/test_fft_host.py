"""The fp32 transform of the spectral screening kernel, run on the host (sgz_fft_reference_host executes the
kernels' own index arithmetic and operations thread by thread), against numpy's fp64 transform.  No GPU."""
import numpy as np
import pytest

from strugatzki_amd import _native as N

lib = N.lib


def _fft(n, inverse, z):
    inp = np.empty(2 * n, dtype=np.float32)
    inp[0::2] = z.real
    inp[1::2] = z.imag
    out = np.empty(2 * n, dtype=np.float32)
    rc = lib.sgz_fft_reference_host(n, int(inverse), N.ptr(inp), N.ptr(out))
    assert rc == 0
    return out[0::2].astype(np.float64) + 1j * out[1::2].astype(np.float64)


@pytest.mark.parametrize("n", [4096, 8192, 16384])
@pytest.mark.parametrize("inverse", [False, True])
def test_transform_matches_numpy(n, inverse):
    rng = np.random.default_rng(n + inverse)
    z = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    got = _fft(n, inverse, z)
    want = np.fft.ifft(z.astype(np.complex128)) * n if inverse else np.fft.fft(z.astype(np.complex128))
    # norm-wise error of an fp32 transform: a few ulp times log2(n), relative to the vector's norm
    err = np.linalg.norm(got - want) / np.linalg.norm(want)
    assert err < 2e-6
    # element-wise against what the kernel's error term assumes of a transform (fft_eta: 40 roundings per pass):
    # an output is a sum of n inputs of modulus-1 weights, so it is within that of sum |z_i| <= sqrt(n) |z|
    passes = 4 if n > 4096 else 3
    bound = passes * 40 * 2.0 ** -24 * np.sqrt(n) * np.linalg.norm(z.astype(np.complex128))
    assert np.max(np.abs(got - want)) <= bound


@pytest.mark.parametrize("n", [4096, 8192, 16384])
def test_impulse_and_single_tone(n):
    z = np.zeros(n, dtype=np.complex64)
    z[5] = 1.0
    got = _fft(n, False, z)
    want = np.exp(-2j * np.pi * 5 * np.arange(n) / n)
    assert np.max(np.abs(got - want)) < 1e-5
    tone = np.exp(2j * np.pi * 37 * np.arange(n) / n).astype(np.complex64)
    got = _fft(n, False, tone)
    assert abs(got[37] - n) < 1e-3 * n
    got[37] = 0
    assert np.max(np.abs(got)) < 2e-3 * np.sqrt(n)


def test_round_trip_and_circular_correlation():
    """Forward, product with a conjugated spectrum, inverse: the sliding sums of a short sequence against a long
    one, as the kernel forms them."""
    n, L = 4096, 700
    rng = np.random.default_rng(3)
    x = rng.standard_normal(n).astype(np.float32)
    a = np.zeros(n, dtype=np.float32)
    a[:L] = rng.standard_normal(L).astype(np.float32)
    X = _fft(n, False, x.astype(np.complex64))
    A = _fft(n, False, a.astype(np.complex64))
    y = _fft(n, True, (np.conj(A) * X / n).astype(np.complex64))
    want = np.array([np.dot(a[:L].astype(np.float64), x[k:k + L].astype(np.float64)) for k in range(n - L + 1)])
    scale = np.linalg.norm(a) * np.linalg.norm(x)
    assert np.max(np.abs(y.real[:n - L + 1] - want)) < 1e-5 * scale
    assert np.max(np.abs(y.imag)) < 1e-5 * scale


def test_argument_check():
    buf = np.zeros(16, dtype=np.float32)
    assert lib.sgz_fft_reference_host(1000, 0, N.ptr(buf), N.ptr(buf)) != 0
    assert lib.sgz_fft_reference_host(4096, 0, None, N.ptr(buf)) != 0
