#!/usr/bin/env python3
"""Every obstacle layout the reference HOLDS: the saved test-environment sets written by the real, visilibity-backed env
(algos/test_environment/eval/test_env_gen.py:13-24, :26-69 -> algos/multiagent/evaluation/test_environments/
test_env_dict_obs<k>_<snr>_v4, k = 1..7, snr in {high, med, low} x 1000 environments, and {none} x 100 for k = 1..6:
21 600 layouts).  Each was accepted by the reference's create_obs (rad_search_env.py:948-1011: rectangles whose boundaries
do not touch), world.is_valid (:788-791) and sample_source_loc_pos (:1013-1131: source / detector outside every rectangle,
>= 1000 cm apart).  Those are one-sided pins for the obstacle rows E3 / E5 / E11 of SURVEY section 8: whatever the reference
accepted, the restated predicates must accept.

Runs in the build container only (the reference tree does not travel); reads the joblib files WITHOUT unpickling
(radiation_ppo_amd.testsets) and writes plain integer arrays:

    python tests/golden/make_testset_layouts.py [/root/reference]   ->  tests/golden/testset_layouts.npz

    k [M] i8, snr [M] i8 (index into SNRS), env [M] i16 (env_<i> of its set), src / det [M, 2] i32, intensity [M] i32, bkg [M] i32,
    rects [M, 7, 4] i32 as (x0, y0, x1, y1), rows >= k zero.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

SNRS = ("none", "low", "med", "high")


def main(ref="/root/reference"):
    from radiation_ppo_amd.testsets import load_test_environments
    root = os.path.join(ref, "algos", "multiagent", "evaluation", "test_environments")
    K, S, E, SRC, DET, I, B, R = [], [], [], [], [], [], [], []
    for k in range(1, 8):
        for si, snr in enumerate(SNRS):
            p = os.path.join(root, f"test_env_dict_obs{k}_{snr}_v4")
            if not os.path.exists(p):
                continue
            sets = load_test_environments(p)
            for q in sorted(sets, key=lambda s: int(s.split("_")[1])):
                src, det, inten, bkg, obstacles = sets[q]
                assert len(obstacles) == k, (p, q, len(obstacles))
                for v in (src, det):
                    assert v.shape == (2,) and np.all(v == np.round(v)), (p, q, v)
                r = np.zeros((7, 4), dtype=np.int32)
                for j, ob in enumerate(obstacles):
                    pts = np.asarray(ob[0], dtype=np.float64)
                    assert pts.shape == (4, 2) and np.all(pts == np.round(pts)), (p, q, pts)
                    x0, y0, x1, y1 = pts[:, 0].min(), pts[:, 1].min(), pts[:, 0].max(), pts[:, 1].max()
                    # the vertex order of create_obs (:975-983): (x0,y0), (x0,y1), (x1,y1), (x1,y0) -- an axis-aligned rectangle
                    assert np.array_equal(pts, [[x0, y0], [x0, y1], [x1, y1], [x1, y0]]), (p, q, pts)
                    r[j] = (x0, y0, x1, y1)
                K.append(k); S.append(si); E.append(int(q.split("_")[1]))
                SRC.append(src.astype(np.int32)); DET.append(det.astype(np.int32)); I.append(int(inten)); B.append(int(bkg)); R.append(r)
            print(f"obs{k}_{snr}: {len(sets)} layouts")
    out = os.path.join(HERE, "testset_layouts.npz")
    np.savez_compressed(out, k=np.array(K, np.int8), snr=np.array(S, np.int8), env=np.array(E, np.int16), src=np.stack(SRC),
                        det=np.stack(DET), intensity=np.array(I, np.int32), bkg=np.array(B, np.int32), rects=np.stack(R))
    print(len(K), "layouts ->", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main(*sys.argv[1:])
