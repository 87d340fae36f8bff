"""Random search configurations (channel counts, target lengths, ragged and short files, weights, boost gates,
numPerFile / minSpacing, screening modes, awkward data) on the GPU against the oracle."""
import os
import sys

import pytest

from common import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12])
def test_random_configurations_match_the_oracle(ctx, seed):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_parity
    assert fuzz_parity.run(120, seed, ctx=ctx) == 0


@pytest.mark.parametrize("seed", [13])
def test_random_configurations_with_either_screening_kernel(ctx, seed):
    """The same sweep with the screening kernel drawn per case (spectral with each transform length, matrix cores,
    the library's choice), and the kernel's upper bounds checked against the oracle's dense similarities."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_parity
    assert fuzz_parity.run(150, seed, ctx=ctx, kernels=True) == 0


@pytest.mark.parametrize("seed", [21])
def test_random_punch_out_searches_match_the_oracle(ctx, seed):
    """Punch-in / punch-out: random spans, gaps, punch ranges, files that the punch-in pass sees shorter than its
    span (one window over the ring's stale contents), constant stretches."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_punch
    assert fuzz_punch.run(150, seed, ctx=ctx) == 0


@pytest.mark.parametrize("seed", [31])
def test_random_segmentation_selfsimilarity_stats_crosssimilarity_inputs_match_the_oracle(ctx, seed):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_halfcorr
    assert fuzz_halfcorr.run(240, seed, ctx=ctx) == 0
