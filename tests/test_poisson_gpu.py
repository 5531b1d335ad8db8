"""E6 on the device: rs_poisson (csrc/rs_device.hpp:104-148) as a distribution, and the integer draws of the reset kernel.
2^20 envs are put on fixed geometry through rs_state_field (detector 1000 cm from the source, so lambda = intensity / 1000 +
bkg with the reference's I / r falloff, rad_search_env.py:498-502) and re-measured in place with the step(None) action; the
2^20 measurements per rate are held to scipy.stats.poisson (chi-square, mean, variance) by the checker of
tests/test_poisson_distribution.py."""
import numpy as np
import pytest
import torch
from scipy import stats

from test_poisson_distribution import LAMBDAS, check_poisson_sample

pytestmark = pytest.mark.gpu


def test_kernel_measurements_are_poisson():
    from radiation_ppo_amd.envs import RadSearchVec
    N = 1 << 20
    vec = RadSearchVec(N, obstruction_count=0, enforce_grid_boundaries=True, seed=12345)
    vec.reset()
    vec.state("src_x")[:] = 1200; vec.state("src_y")[:] = 1200
    vec.state("x")[:] = 200; vec.state("y")[:] = 1200
    vec.state("sp")[:] = 1000.0; vec.state("prev")[:] = 1000.0
    none = torch.full((N, 1), 9, dtype=torch.int8, device="cuda")          # step(None): no move, a fresh measurement
    for lam in LAMBDAS:
        vec.state("intensity")[:] = int(round(lam * 1000)); vec.state("bkg")[:] = 0
        obs, *_ = vec.step(none)
        x = obs[:, 0, 0].double().cpu().numpy()
        assert np.array_equal(x, np.rint(x))
        check_poisson_sample(x.astype(np.int64), lam, "kernel")
    # background added to the rate; blocked line of sight is exercised by the parity tests
    vec.state("intensity")[:] = 2_000_000; vec.state("bkg")[:] = 37
    obs, *_ = vec.step(none)
    check_poisson_sample(obs[:, 0, 0].double().cpu().numpy().astype(np.int64), 2037.0, "kernel+bkg")
    assert vec.error_flags() == 0


def test_reset_integer_draws_are_uniform():
    """RsDrawSeq::integers in the reset kernel: detector cells (drawn unconditionally, rad_search_env.py:1032-1035),
    intensities [1e6, 1e7) and backgrounds [10, 51) (:778-779) of 2^20 fresh envs are uniform on their ranges."""
    from radiation_ppo_amd.envs import RadSearchVec
    N = 1 << 20
    vec = RadSearchVec(N, obstruction_count=0, enforce_grid_boundaries=True, seed=999)
    vec.reset()
    for name, lo, hi, bins in (("x", 200, 2200, 400), ("y", 200, 2200, 400), ("intensity", 1_000_000, 10_000_000, 500),
                               ("bkg", 10, 51, 41)):
        v = vec.state(name).cpu().numpy().reshape(-1).astype(np.int64)
        assert v.min() >= lo and v.max() < hi
        cnt = np.bincount((v - lo) * bins // (hi - lo), minlength=bins)
        chi2 = ((cnt - N / bins) ** 2 / (N / bins)).sum()
        assert stats.chi2.sf(chi2, bins - 1) > 1e-4, (name, chi2)
    # joint uniformity of the detector cell on a 20 x 20 grid (the two coordinates come from consecutive draws)
    x = vec.state("x").cpu().numpy().reshape(-1).astype(np.int64); y = vec.state("y").cpu().numpy().reshape(-1).astype(np.int64)
    cnt = np.bincount(((x - 200) // 100) * 20 + (y - 200) // 100, minlength=400)
    chi2 = ((cnt - N / 400) ** 2 / (N / 400)).sum()
    assert stats.chi2.sf(chi2, 399) > 1e-4, chi2
