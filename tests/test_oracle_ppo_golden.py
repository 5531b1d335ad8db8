"""PPO-side oracle functions vs golden vectors from the reference (ppo.py PPOBuffer, RADTEAM_core
StatisticStandardization) and vs the known-answer vectors of the reference's own unit tests."""
import os

import numpy as np

from oracle.radsearch_oracle import WelfordOracle, discount_cumsum, gae_and_rtg


def test_gae_rtg_matches_reference_buffer(golden_dir):
    g = dict(np.load(os.path.join(golden_dir, "gae.npz")).items())
    rew = g["rew"].astype(np.float32)      # PPOBuffer stores float32 (ppo.py:323-325)
    val = g["val"].astype(np.float32)
    gamma, lam = float(g["gamma"]), float(g["lam"])
    T = len(rew)
    adv = np.zeros(T, np.float32)
    ret = np.zeros(T, np.float32)
    start = 0
    for t in range(T):
        if g["cut"][t]:
            a, r = gae_and_rtg(rew[start:t + 1], val[start:t + 1], g["last_val"][t], gamma, lam)
            adv[start:t + 1] = np.array(a, dtype=np.float64).astype(np.float32)
            ret[start:t + 1] = np.array(r, dtype=np.float64).astype(np.float32)
            start = t + 1
    assert start == T
    assert np.array_equal(adv, g["adv_raw"])      # bit-exact: same float64 recurrence, same rounding
    assert np.array_equal(ret, g["ret"])
    # advantage normalisation (ppo.py:445-446 via mpi_tools.py:71-95, 1 rank): float32 mean / population std
    x = adv
    mean = np.sum(x, dtype=np.float32) / np.float32(len(x))
    std = np.sqrt(np.sum((x - mean) ** 2, dtype=np.float32) / np.float32(len(x)))
    assert np.allclose((x - mean) / std, g["adv_norm"], rtol=1e-5, atol=1e-6)


def test_reference_unit_test_vectors():
    """unit_tests/test_PPO.py:463-497 known answers (values copied as data, not code)."""
    rewards = [-0.46, -0.48, -0.46, -0.45, -0.45, -0.46, -0.47, -0.49, -0.49, -0.5, -0.5]
    values = [-0.26629043, -0.26634163, -0.26633365, -0.26649433, -0.26669, -0.2670981, -0.26757, -0.26831, -0.26877,
              -0.26949, -0.26949]
    # discount_cumsum identity used by the reference's own Helpers (test_PPO.py:154-215)
    y = discount_cumsum(rewards, 0.99)
    acc = 0.0
    for t in reversed(range(len(rewards))):
        acc = rewards[t] + 0.99 * acc
        assert y[t] == acc
    adv, rtg = gae_and_rtg(np.float32(rewards[:-1]), np.float32(values[:-1]), values[-1], 0.99, 0.9)
    assert len(adv) == 10 and len(rtg) == 10
    # loop GAE of the reference (ppo.py:89-124) on the same data agrees to round-off
    last_adv, last_val = 0.0, float(np.float32(values[-2]))
    vals = [float(v) for v in np.float32(values[:-1])] + [values[-1]]
    rews = [float(r) for r in np.float32(rewards[:-1])]
    for t in reversed(range(10)):
        delta = rews[t] + 0.99 * vals[t + 1] - vals[t]
        last_adv = delta + 0.99 * 0.9 * last_adv
        assert abs(adv[t] - last_adv) < 1e-12


def test_welford_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "welford.npz"))
    st = WelfordOracle()
    for x, z in zip(g["x"], g["z"]):
        st.update(float(x))
        assert st.standardize(float(x)) == z
    # reference's own known answer (unit_tests/test_RADTEAM_core.py:144-236): 1000, 2000, 100
    st = WelfordOracle()
    for x in (1000.0, 2000.0, 100.0):
        st.update(x)
    assert abs(st.mean - 1033.3333333333333) < 1e-9
    assert abs(st.std - 950.4384952922168) < 1e-9
    assert abs(st.standardize(100.0) - (-0.9820028733646521)) < 1e-12
