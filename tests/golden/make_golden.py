#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the CPU oracle.

The reference (Scala/JVM) cannot be run in this environment and ships no
numeric fixtures, so these vectors are outputs of oracle/strugatzki_oracle.c on
seeded synthetic inputs; they freeze the oracle's behaviour (any later change
to it shows up as a diff) and give the HIP kernels fixed inputs / outputs to be
checked against.  Run from the repository root: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from common import (NUM_CH, SEED, STEP, O, corr_config, make_plants, norms_default, punch_config, synth_db,  # noqa: E402
                    synth_target)


def corr_fixture():
    L, ts, nf, frames = 32, 12, 3, 400
    plants = make_plants(2, L, nf, frames, ts, amp_step=0.05)
    db = synth_db(nf, frames, plants)
    tgt = synth_target(ts, L)
    norms = norms_default()
    out = dict(db=np.stack(db), target=tgt, norms=norms, L=L, ts=ts)
    settings = [(4, 1, 0), (5, 2, 0), (6, 3, 20 * STEP)]
    for k, (nm, npf, ms) in enumerate(settings):
        cfg = corr_config(ts, L, num_matches=nm, num_per_file=npf, min_spacing=ms)
        m, sims, boosts = O.correlation_run(cfg, tgt, norms, db)
        out[f"matches{k}"] = np.array([O.match_tuple(x) for x in m], dtype=np.float64)
        out[f"setting{k}"] = np.array([nm, npf, ms])
    out["sims"] = np.stack(sims)
    out["boosts"] = np.stack(boosts)
    # a second pass without normalisation and with temporal weight 0.2
    cfg = corr_config(ts, L, weight=0.2, normalize=False, num_matches=3)
    m, sims, boosts = O.correlation_run(cfg, tgt, None, db)
    out["sims_raw_w02"] = np.stack(sims)
    np.savez_compressed(os.path.join(HERE, "correlation.npz"), **out)


PUNCH = dict(Lin=40, Lout=24, ts=12, gap=16, nf=3, frames=360,
             settings=[(4, 1, 0, 10, 40), (5, 2, 20 * STEP, 0, 8)])  # numMatches, numPerFile, minSpacing, min/maxPunch


def punch_fixture():
    q = PUNCH
    span = q["Lin"] + q["gap"] + q["Lout"]
    db = synth_db(q["nf"], q["frames"], make_plants(2, span, q["nf"], q["frames"], q["ts"], amp_step=0.05))
    tgt = synth_target(q["ts"], span)
    norms = norms_default()
    out = dict(db=np.stack(db), target=tgt, norms=norms)
    for k, (nm, npf, ms, lo, hi) in enumerate(q["settings"]):
        cfg = punch_config(q["ts"], q["Lin"], q["ts"] + q["Lin"] + q["gap"], q["Lout"], min_punch=lo, max_punch=hi,
                           out_weight=0.25, num_matches=nm, num_per_file=npf, min_spacing=ms)
        m, din, dout = O.correlation_run_punch(cfg, tgt, norms, db)
        out[f"matches{k}"] = np.array([O.match_tuple(x) for x in m], dtype=np.float64)
        out[f"dense_in{k}"] = np.stack(din)
        out[f"dense_out{k}"] = np.stack(dout)
    np.savez_compressed(os.path.join(HERE, "punch.npz"), **out)


def segm_fixture():
    frames = np.concatenate([O.synth_file(SEED, k, 150, NUM_CH) * np.float32(1.0 - 0.3 * (k % 2)) for k in range(4)])
    norms = norms_default()
    cfg = O.SegmConfig(1, 1, 20 * STEP, 560 * STEP, 22050, 0.5, 1, 5, 22050, STEP, NUM_CH - 1)
    breaks, dense = O.segmentation_run(cfg, frames, norms)
    np.savez_compressed(os.path.join(HERE, "segmentation.npz"), frames=frames, norms=norms, dense=dense,
                        breaks=np.array([(b.sim, b.pos) for b in breaks], dtype=np.float64),
                        params=np.array([20, 560, 43, 5, 22050]))


def selfsim_fixture():
    frames = O.synth_file(SEED, 5, 210, NUM_CH)
    frames[100:140] = frames[20:60]  # a repeated passage -> an off-diagonal stripe
    norms = norms_default()
    cfg = O.SelfSimConfig(0, 0, 0, 0, 8 * STEP, 3, 0.5, 0, 0.7, 0.9, 0, 1, STEP, NUM_CH - 1)
    ext, decim, af_start, num_corrs = O.selfsim_geometry(cfg, frames.shape[0], frames.shape[0])
    sim, img = O.selfsim_run(cfg, frames, None, norms)
    np.savez_compressed(os.path.join(HERE, "selfsim.npz"), frames=frames, norms=norms, sim=sim, img=img,
                        params=np.array([ext, decim, af_start, num_corrs, 8]))


if __name__ == "__main__":
    corr_fixture()
    punch_fixture()
    segm_fixture()
    selfsim_fixture()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
