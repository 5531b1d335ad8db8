"""E6: the measurement sampler as a DISTRIBUTION (the reference draws numpy's Generator.poisson, rad_search_env.py:498-502).

The oracle's `poisson_from_uniforms` and the kernel's `rs_poisson` (csrc/rs_device.hpp) share one rewrite of Hormann's
PTRS acceptance test (one log + a Stirling series), so comparing them with each other cannot see a mistake in it.  Here
both are held to the Poisson law itself: chi-square against scipy.stats.poisson.pmf and mean / variance within 4 sigma,
for rates that cover every branch -- the multiplication method (< 10), the 9.99 / 10 switch, the squeeze-accept fast path,
the k < 0 rejection, the lgamma branch (k + 1 < 10) and the Stirling branch of the slow path, up to the largest rate the
env produces (1e7 / 110 + 50 ~ 9e4).  Also the integer draws (RsDrawSeq::integers / PhiloxDraws.integers) for uniformity."""
import numpy as np
import pytest
from scipy import stats

from oracle.radsearch_oracle import PhiloxDraws, poisson_from_uniforms

LAMBDAS = [0.3, 3.0, 9.99, 10.0, 45.0, 500.0, 5.0e3, 9.0e4]


def check_poisson_sample(x: np.ndarray, lam: float, tag=""):
    """chi-square goodness of fit (bins with expected count >= 20, tails pooled) at p > 1e-4; mean and variance within
    4 standard errors (Var[sample variance] = (mu4 - sigma^4) / n with mu4 = lam + 3 lam^2 for Poisson)."""
    n = x.size
    assert x.min() >= 0
    mean, var = x.mean(), x.var()
    assert abs(mean - lam) <= 4.0 * np.sqrt(lam / n), (tag, lam, "mean", mean)
    se_var = np.sqrt((lam + 3 * lam * lam - lam * lam) / n)
    assert abs(var - lam) <= 4.0 * se_var + lam / n, (tag, lam, "var", var)
    lo = int(max(0, np.floor(lam - 8 * np.sqrt(lam) - 8)))
    hi = int(np.ceil(lam + 8 * np.sqrt(lam) + 12))
    ks = np.arange(lo, hi + 1)
    pmf = stats.poisson.pmf(ks, lam)
    pmf[0] += stats.poisson.cdf(lo - 1, lam)
    pmf[-1] += stats.poisson.sf(hi, lam)
    cnt = np.bincount(np.clip(x, lo, hi).astype(np.int64) - lo, minlength=ks.size).astype(np.float64)
    exp = pmf * n
    # pool neighbouring bins until every expected count is >= 20
    oe, ee, o_acc, e_acc = [], [], 0.0, 0.0
    for o, e in zip(cnt, exp):
        o_acc += o; e_acc += e
        if e_acc >= 20:
            oe.append(o_acc); ee.append(e_acc); o_acc = e_acc = 0.0
    if e_acc > 0 and ee:
        oe[-1] += o_acc; ee[-1] += e_acc
    oe, ee = np.array(oe), np.array(ee)
    chi2 = ((oe - ee) ** 2 / ee).sum()
    p = stats.chi2.sf(chi2, len(ee) - 1)
    assert p > 1e-4, (tag, lam, "chi2", chi2, "dof", len(ee) - 1, "p", p)


def test_checker_has_power():
    """The checker itself: accepts numpy's Poisson, rejects a sampler whose mean is 8 standard errors high (a +1 on a
    fraction of the draws) and one whose variance is 3 % too wide."""
    rng = np.random.default_rng(5)
    for lam in (3.0, 45.0, 5.0e3):
        x = rng.poisson(lam, 400_000)
        check_poisson_sample(x, lam)
        bad = x + (rng.random(x.size) < min(1.0, 8.0 * np.sqrt(lam / x.size)))
        with pytest.raises(AssertionError):
            check_poisson_sample(bad, lam)
    wide = np.rint(rng.normal(5.0e3, np.sqrt(5.0e3) * 1.03, 400_000))
    with pytest.raises(AssertionError):
        check_poisson_sample(wide, 5.0e3)


@pytest.mark.parametrize("lam", LAMBDAS)
def test_oracle_sampler_is_poisson(lam):
    n = 1_000_000 if lam >= 10 else 300_000       # the multiplication method loops ~lam times per draw in pure Python
    rng = np.random.default_rng(int(lam * 100) + 1)
    block = rng.random((2, 4 * n))
    pos = [0]

    def next_uv(i):
        j = pos[0]
        pos[0] = j + 1
        return block[0, j], block[1, j]
    out = np.empty(n, dtype=np.int64)
    for q in range(n):
        out[q] = poisson_from_uniforms(lam, next_uv)
        if pos[0] > block.shape[1] - 64:
            block = rng.random((2, 4 * n)); pos[0] = 0
    check_poisson_sample(out, lam, "oracle")


def test_philox_measurement_stream_is_poisson():
    """The same through the REAL draw source: PhiloxDraws.poisson with its (attempt, step, episode, stream) counters, over
    env ids and steps -- the uniforms of the Philox stream are what the kernel consumes."""
    lam = 37.5
    out = []
    for env in range(40):
        d = PhiloxDraws(289714752, env)
        d.begin_reset(env % 3)
        for t in range(500):
            d.begin_step(t)
            out.append(d.poisson(lam, env % 2))
    check_poisson_sample(np.array(out), lam, "philox")


def test_integer_draws_are_uniform():
    """PhiloxDraws.integers (= RsDrawSeq::integers, Lemire's multiply-shift on 64 random bits): chi-square uniformity on
    the ranges the reset uses -- [200, 2200) coordinates, [1e6, 1e7) intensities (binned), [10, 51) backgrounds, [1, 6)."""
    for lo, hi, bins in ((200, 2200, 200), (1_000_000, 10_000_000, 300), (10, 51, 41), (1, 6, 5)):
        vals = []
        for env in range(60):
            d = PhiloxDraws(77, env)
            d.begin_reset(env)
            vals += [d.integers(lo, hi) for _ in range(1000)]
        v = np.array(vals)
        assert v.min() >= lo and v.max() < hi
        cnt = np.bincount(((v - lo) * bins // (hi - lo)).astype(np.int64), minlength=bins)
        chi2 = ((cnt - v.size / bins) ** 2 / (v.size / bins)).sum()
        assert stats.chi2.sf(chi2, bins - 1) > 1e-4, (lo, hi, chi2)
