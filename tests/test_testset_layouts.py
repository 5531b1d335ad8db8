"""One-sided pins of the obstacle rows (E3, E5, E11) from data the reference HOLDS: the 21 600 obstacle layouts of
its saved test-environment sets (tests/golden/testset_layouts.npz, made by tests/golden/make_testset_layouts.py from
algos/multiagent/evaluation/test_environments/test_env_dict_obs{1..7}_*_v4, written by the real visilibity-backed env through
algos/test_environment/eval/test_env_gen.py:13-69).  Every one of them was ACCEPTED by the reference's create_obs (rad_search_env.py
:948-1011, the boundary_distance test :988), world.is_valid (:788-791) and sample_source_loc_pos (:1013-1131: Point._in :1061,
:1108; the 1000 cm rule), so the restated predicates of oracle/radsearch_oracle.py must accept every one of them too; and the
unconditioned `none` sets give the distribution the restated reset has to reproduce (line-of-sight-blocked starts :1105-1123)."""
import math
import os

import numpy as np
import pytest
from scipy import stats

from oracle.radsearch_oracle import (PhiloxDraws, RadSearchOracle, layout_is_valid, pt_in_closed, pt_in_closed_eps,
                                     seg_rect_boundary_lt_1e3)

SNRS = ("none", "low", "med", "high")
EPSILON = 1e-7          # rad_search_env.py:64


@pytest.fixture(scope="module")
def sets(golden_dir):
    z = np.load(os.path.join(golden_dir, "testset_layouts.npz"))
    return {k: z[k] for k in z.files}


def test_fixture_is_the_whole_of_the_reference_s_sets(sets):
    k, snr = sets["k"], sets["snr"]
    assert len(k) == 21600
    for kk in range(1, 8):
        for si, name in enumerate(SNRS):
            want = 0 if (name == "none" and kk == 7) else (100 if name == "none" else 1000)
            assert int(np.sum((k == kk) & (snr == si))) == want, (kk, name)
    # the draw ranges of create_obs (:961-973) and of reset (:778-779), attained at both ends over 86 800 rectangles
    r = sets["rects"]
    m = np.arange(7)[None, :] < k[:, None]
    assert not r[~m].any()
    for c, lo, hi in ((r[..., 0][m], 200, 1979), (r[..., 1][m], 200, 1979), ((r[..., 2] - r[..., 0])[m], 200, 499),
                      ((r[..., 3] - r[..., 1])[m], 200, 499)):
        assert c.min() == lo and c.max() == hi
    for p in (sets["src"], sets["det"]):
        assert p.min() == 200 and p.max() == 2199                      # integers(200, 2200) on both axes (:1032-1035)
    assert 1_000_000 <= sets["intensity"].min() and sets["intensity"].max() < 10_000_000
    assert sets["bkg"].min() == 10 and sets["bkg"].max() == 50


def test_every_saved_layout_passes_the_restated_acceptance_predicates(sets):
    """What the reference accepted, the restatement accepts: no pair of rectangles with touching boundaries (:988), the layout
    is_valid (:788), source and detector outside every rectangle even with visilibity's epsilon (:1061, :1108), >= 1000 cm apart."""
    bad = []
    for i in range(len(sets["k"])):
        k = int(sets["k"][i])
        rects = [tuple(int(v) for v in sets["rects"][i, j]) for j in range(k)]
        sx, sy = (int(v) for v in sets["src"][i])
        dx, dy = (int(v) for v in sets["det"][i])
        ok = layout_is_valid(rects)
        for a in range(k):
            for b in range(a):
                ok = ok and not RadSearchOracle._rect_boundaries_touch(rects[b], rects[a])
            for (px, py) in ((sx, sy), (dx, dy)):
                ok = ok and not pt_in_closed(px, py, rects[a]) and not pt_in_closed_eps(float(px), float(py), rects[a], EPSILON)
        ok = ok and math.sqrt(float((sx - dx) ** 2 + (sy - dy) ** 2)) >= 1000.0
        if not ok:
            bad.append(i)
    assert not bad, bad[:10]


def _blocked(rects, s, d):
    return any(seg_rect_boundary_lt_1e3(int(d[0]), int(d[1]), int(s[0]), int(s[1]), r) for r in rects)


def test_snr_sets_are_what_their_name_says(sets):
    """test_env_gen.py:37-49: a set keeps the environments whose start has (I / d^2 + bkg) / bkg in its band -- Euclidean distance,
    inverse SQUARE -- which checks that the fixture's intensity / background / coordinates belong together."""
    d2 = ((sets["src"].astype(np.float64) - sets["det"]) ** 2).sum(1)
    snr = (sets["intensity"] / d2 + sets["bkg"]) / sets["bkg"]
    for si, (lo, hi) in ((1, (1.0, 1.2)), (2, (1.2, 1.6)), (3, (1.6, 2.0))):
        m = sets["snr"] == si
        assert np.all((snr[m] > lo) & (snr[m] <= hi)), SNRS[si]


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 6])
def test_restated_reset_reproduces_the_start_distribution_of_the_none_sets(sets, k):
    """The `none` sets are unconditioned resets of the reference at obstruction_count = k (100 each).  The oracle's reset at the same
    count must give (i) the same fraction of starts whose line of sight is blocked (the <= 20 resampling rounds of :1113-1123 make
    that fraction rise with k: 0.8 / 0.93 / 1.0 ... in the saved sets), within binomial error of the 100 saved starts, and (ii)
    rectangle seeds / extents and start distances that a two-sample Kolmogorov-Smirnov test cannot tell from the saved ones."""
    m = (sets["k"] == k) & (sets["snr"] == 0)
    saved = [_blocked([tuple(int(v) for v in r) for r in sets["rects"][i, :k]], sets["src"][i], sets["det"][i]) for i in np.nonzero(m)[0]]
    n_saved, p_saved = len(saved), float(np.mean(saved))
    n_or = 400
    env = RadSearchOracle(PhiloxDraws(20231, k), number_agents=1, obstruction_count=k, enforce_grid_boundaries=True)
    got, seeds, exts, dist = [], [], [], []
    for _ in range(n_or):
        env.epoch_end = True
        env.reset()
        assert env.err == 0
        got.append(_blocked(env.rects, env.src, env.agents[0].det))
        seeds += [r[0] for r in env.rects] + [r[1] for r in env.rects]
        exts += [r[2] - r[0] for r in env.rects] + [r[3] - r[1] for r in env.rects]
        dist.append(math.dist(env.src, env.agents[0].det))
    p = float(np.mean(got))
    se = math.sqrt(max(p * (1 - p), 0.01) * (1 / n_saved + 1 / n_or))
    assert abs(p - p_saved) <= 4 * se, (k, p, p_saved, se)
    # rectangles: every saved set of this k (the SNR bands condition on distance and intensity, not on the layout)
    mk = sets["k"] == k
    r = sets["rects"][mk][:, :k]
    assert stats.ks_2samp(np.concatenate([r[..., 0].ravel(), r[..., 1].ravel()]), seeds).pvalue > 1e-3
    assert stats.ks_2samp(np.concatenate([(r[..., 2] - r[..., 0]).ravel(), (r[..., 3] - r[..., 1]).ravel()]), exts).pvalue > 1e-3
    d_saved = np.linalg.norm(sets["src"][m].astype(np.float64) - sets["det"][m], axis=1)
    assert stats.ks_2samp(d_saved, dist).pvalue > 1e-3
