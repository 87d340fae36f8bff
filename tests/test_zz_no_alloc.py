"""sgz_corr_rescore_device_noalloc (re-scoring with the partial sums in the library's fixed device array, no
allocation) against sgz_corr_rescore_device, through the C ABI: bit-identical records, on two calls in a row,
the second with a smaller input; sizes beyond the array are refused."""
import numpy as np
import pytest

from common import NUM_CH, STEP, HipMem, O, corr_config, make_plants, norms_default, synth_db, synth_target

pytestmark = pytest.mark.gpu


def _device_list(hip, cap):
    rec, cnt = hip.alloc(16 * cap), hip.alloc(8)
    hip.upload(rec, np.zeros((cap, 4), np.int32))
    hip.upload(cnt, np.zeros(1, np.int64))
    return rec, cnt


def _candidates(ctx, hip, nf, want, cap):
    """The `want` windows with the largest screened similarities (or a few more), as device records + count,
    in (file, offset) order."""
    dense = np.concatenate([ctx.corr_read_dense(f)[0] for f in range(nf)])
    thr = float(np.sort(dense)[-want - 1])
    rec, cnt = _device_list(hip, cap)
    ctx.corr_compact_device(0, nf, thr, rec, cap, cnt)
    ctx.synchronize()
    n = int(hip.download(cnt, np.zeros(1, np.int64))[0])
    assert 0 < n <= cap
    h = hip.download(rec, np.zeros((cap, 4), np.int32))[:n]
    h = h[np.lexsort((h[:, 1], h[:, 0]))]
    hip.upload(rec, h)
    return rec, cnt, n, h


def test_rescore_noalloc_equals_rescore(ctx):
    import strugatzki_amd as S
    from strugatzki_amd import _native as N

    hip = HipMem()
    ts, norms = 9, norms_default()
    try:
        # first call: long target (several slices per window), many candidates; second: shorter target, fewer
        for L, nf, frames, want, cap in ((2584, 3, 6000, 700, 8192), (200, 2, 1500, 90, 1024)):
            db = synth_db(nf, frames, make_plants(2, L, nf, frames, ts))
            tgt = synth_target(ts, L)
            ctx.db_reset(NUM_CH)
            for f in db:
                ctx.db_add_file(f)
            ctx.set_norms(norms)
            ctx.corr_set_target(tgt[ts:ts + L], 0.5, 8.0)
            _, sims, boosts = O.correlation_run(corr_config(ts, L, num_matches=3), tgt, norms, db)
            ctx.corr_run_screen(0, nf)  # fp32-product screening values: the re-scoring has something to correct
            rec_a, cnt_a, n, h = _candidates(ctx, hip, nf, want, cap)
            rec_b, cnt_b = _device_list(hip, cap)
            hip.upload(rec_b, h)
            hip.upload(cnt_b, np.array([n], np.int64))
            ctx.corr_rescore_device(rec_a, cap, cnt_a)
            ctx.synchronize()
            free_before = hip.free_bytes()
            ctx.corr_rescore_device_noalloc(rec_b, cap, cnt_b)
            ctx.synchronize()
            assert hip.free_bytes() == free_before  # the call took no device memory
            a = hip.download(rec_a, np.zeros((cap, 4), np.int32))[:n]
            b = hip.download(rec_b, np.zeros((cap, 4), np.int32))[:n]
            assert n > 20 and np.array_equal(a, b)  # same kernels, same order of operations: identical bits
            assert not np.array_equal(b, h)         # (the screened values were replaced)
            got = np.ascontiguousarray(b).view(N.CANDIDATE_DTYPE).reshape(-1)
            for c in got:  # and they are the exact values
                assert abs(c["sim"] - sims[c["file"]][c["offset"]]) <= 1.2e-7
                assert abs(c["boost"] - boosts[c["file"]][c["offset"]]) <= 2e-6 * boosts[c["file"]][c["offset"]]
        # beyond the array: refused with a clear message, nothing launched
        rec, cnt = _device_list(hip, 4)
        with pytest.raises(S.NativeError, match="scratch") as e:
            ctx.corr_rescore_device_noalloc(rec, 1 << 20, cnt)
        assert e.value.code == N.SGZ_E_ARG
        with pytest.raises(S.NativeError, match="scratch") as e:  # (a size whose byte count does not fit 64 bits)
            ctx.corr_rescore_device_noalloc(rec, (1 << 62) + 5, cnt)
        assert e.value.code == N.SGZ_E_ARG
    finally:
        ctx.synchronize()
        hip.free_all()
