"""the batched PFGRUCell (radiation_ppo_amd/pfgru.py) against the reference's own cell
(algos/test_cnn/RADTEAM_core.py:1418-1666) with every random draw the reference made replayed (tests/golden/pfgru.npz):
location predictions, resampled particles and log weights, step by step, for a carried hidden state and for the CNN harness'
"every step from h0" usage; state_dict keys interchange.  Plus the counter-based draw source: uniformity / normality and
independence of the sharding."""
import os

import numpy as np
import torch
from scipy import stats

from radiation_ppo_amd.pfgru import PFGRUCell, PredictorBank, hash_normal, hash_uniform


def _cell(g):
    cell = PFGRUCell(input_size=3, obs_size=3, hidden_size=24)
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd_")}
    assert sorted(sd) == sorted(cell.state_dict())                  # predictor.pt written by either side loads in the other
    cell.load_state_dict(sd)
    return cell.eval()


def test_cell_matches_reference_step_by_step(golden_dir):
    g = np.load(os.path.join(golden_dir, "pfgru.npz"))
    cell = _cell(g)
    obs = torch.from_numpy(g["obs"])
    with torch.no_grad():
        for tag, carry in (("carry", True), ("fresh", False)):
            h, p = cell.init_hidden(1, u=torch.from_numpy(g[f"{tag}_h0"]).unsqueeze(0), device="cpu")
            for t in range(obs.shape[0]):
                loc, (h1, p1) = cell(obs[t:t + 1], (h, p), torch.from_numpy(g[f"{tag}_eps"][t]).unsqueeze(0),
                                     resample_idx=torch.from_numpy(g[f"{tag}_idx"][t]).unsqueeze(0))
                assert np.allclose(loc[0].numpy(), g[f"{tag}_loc"][t].reshape(-1), rtol=1e-5, atol=1e-6), (tag, t)
                assert np.allclose(h1[0].numpy(), g[f"{tag}_h"][t], rtol=1e-5, atol=1e-6), (tag, t)
                assert np.allclose(p1[0].numpy(), g[f"{tag}_p"][t], rtol=1e-5, atol=2e-6), (tag, t)
                if carry:
                    h, p = h1, p1
    assert (g["carry_loc"] >= 0).all()                               # the ReLU behind the last layer (RADTEAM_core.py:1574-1577)


def test_batched_equals_per_sample_and_inverse_cdf_resampling(golden_dir):
    """B independent problems in one call equal B single calls; resampling from uniforms follows the soft-resampling
    distribution alpha * w + (1 - alpha) / P (chi-square over many draws)."""
    g = np.load(os.path.join(golden_dir, "pfgru.npz"))
    cell = _cell(g)
    rng = np.random.default_rng(3)
    B, P, H = 5, 40, 24
    obs = torch.from_numpy(rng.uniform(0, 2, (B, 3)).astype(np.float32))
    h0 = torch.from_numpy(rng.random((B, P, H)).astype(np.float32))
    p0 = torch.log_softmax(torch.from_numpy(rng.normal(size=(B, P)).astype(np.float32)), dim=1)
    eps = torch.from_numpy(rng.normal(size=(B, P, H)).astype(np.float32))
    ru = torch.from_numpy(rng.random((B, P)))
    with torch.no_grad():
        loc, (h1, p1) = cell(obs, (h0, p0), eps, resample_u=ru)
        for b in range(B):
            lb, (hb, pb) = cell(obs[b:b + 1], (h0[b:b + 1], p0[b:b + 1]), eps[b:b + 1], resample_u=ru[b:b + 1])
            assert torch.allclose(lb[0], loc[b], atol=1e-6) and torch.allclose(hb[0], h1[b], atol=1e-6) and torch.allclose(pb[0], p1[b], atol=1e-6)
        # distribution of the resampled indices for one fixed weight vector
        cell2 = PFGRUCell(hidden_size=H)
        cell2.load_state_dict(cell.state_dict())
        n = 4000
        big_u = torch.from_numpy(rng.random((n, P)))
        _, (hh, _) = cell(obs[:1].expand(n, 3), (h0[:1].expand(n, P, H).contiguous(), p0[:1].expand(n, P).contiguous()),
                          eps[:1].expand(n, P, H).contiguous(), resample_u=big_u)
        # recover which particle was chosen by matching rows of the pre-resampling particles
        _, (h_pre, p_pre) = PFGRUCell.forward(_NoResample(cell), obs[:1], (h0[:1], p0[:1]), eps[:1])
        w = 0.7 * torch.exp(p_pre[0]).double() + 0.3 / P
        w = (w / w.sum()).numpy()
        d = ((hh.unsqueeze(2) - h_pre[0].view(1, 1, P, H)) ** 2).sum(-1)            # [n, P, P]
        chosen = d.argmin(-1).reshape(-1).numpy()
        cnt = np.bincount(chosen, minlength=P)
        chi2 = ((cnt - w * chosen.size) ** 2 / (w * chosen.size)).sum()
        assert stats.chi2.sf(chi2, P - 1) > 1e-4, chi2


class _NoResample:
    """View of a cell with use_resampling off (the particles before the resampling step)."""

    def __init__(self, cell):
        self.__dict__.update(cell.__dict__)
        self.use_resampling = False
        self._cell = cell

    def __getattr__(self, k):
        return getattr(self._cell, k)


def test_hash_uniform_statistics():
    u = hash_uniform(torch.arange(1 << 20, dtype=torch.int64) * 7919 + 12345).numpy()
    assert u.min() >= 0.0 and u.max() < 1.0
    cnt = np.bincount((u * 256).astype(np.int64), minlength=256)
    assert stats.chi2.sf(((cnt - u.size / 256) ** 2 / (u.size / 256)).sum(), 255) > 1e-4
    assert abs(np.corrcoef(u[:-1], u[1:])[0, 1]) < 5e-3              # consecutive keys are uncorrelated
    u2 = hash_uniform(torch.arange(1 << 20, dtype=torch.int64) * 7919 + 12345 + 2048).numpy()
    z = np.sqrt(-2 * np.log(1 - u)) * np.cos(2 * np.pi * u2)         # the Box-Muller normals of PredictorBank.predict
    assert abs(z.mean()) < 5e-3 and abs(z.std() - 1) < 5e-3 and stats.kstest(z[:200000], "norm").pvalue > 1e-4


def test_predictor_bank_is_sharding_invariant_and_deterministic():
    """Draws are keyed by the GLOBAL env id: envs 8..15 of a 16-env bank equal the 8 envs of a bank with env_id_base = 8."""
    torch.manual_seed(0)
    full = PredictorBank(16, 2, seed=5, env_id_base=0, device="cpu", impl="torch")
    half = PredictorBank(8, 2, seed=5, env_id_base=8, device="cpu", impl="torch")
    for a in range(2):
        half.load_state_dict(a, full.state_dict(a))
    rng = np.random.default_rng(1)
    obs = torch.from_numpy(rng.uniform(0, 1, (16, 2, 11)).astype(np.float32))
    full.reset(); half.reset()
    assert torch.equal(full.h[:, 8:], half.h)
    for t in range(3):
        pf, ph = full.predict(obs), half.predict(obs[8:])
        # identical draws (checked above on h0); the Linear layers may block differently for 16 and 8 rows on the host BLAS
        assert torch.allclose(pf[8:], ph, rtol=1e-5, atol=1e-6) and torch.isfinite(pf).all() and (pf >= 0).all()
        if t == 1:                                                   # some envs start a new episode: fresh h0, new draw keys
            m = torch.arange(16) % 3 == 0
            full.reset(m); half.reset(m[8:])
            assert torch.equal(full.h[:, 8:], half.h)
    again = PredictorBank(16, 2, seed=5, env_id_base=0, device="cpu", impl="torch")
    for a in range(2):
        again.load_state_dict(a, full.state_dict(a))
    again.reset()
    first = again.predict(obs)
    other = PredictorBank(16, 2, seed=6, env_id_base=0, device="cpu", impl="torch")
    for a in range(2):
        other.load_state_dict(a, full.state_dict(a))
    other.reset()
    assert not torch.equal(first, other.predict(obs))                # a different seed gives different draws
    again2 = PredictorBank(16, 2, seed=5, env_id_base=0, device="cpu", impl="torch")
    for a in range(2):
        again2.load_state_dict(a, full.state_dict(a))
    again2.reset()
    assert torch.equal(first, again2.predict(obs))                   # the same seed reproduces them


def test_bank_batched_over_owners_equals_the_cell():
    """PredictorBank.predict runs all owners through stacked matrix products; owner by owner it must equal PFGRUCell.forward
    (the function pinned to the reference above) on the same draws."""
    import math
    torch.manual_seed(3)
    bank = PredictorBank(6, 3, seed=11, env_id_base=40, device="cpu", impl="torch")
    rng = np.random.default_rng(2)
    obs = torch.from_numpy(rng.uniform(0, 1.5, (6, 3, 11)).astype(np.float32))
    bank.reset()
    h0, p0 = bank.h.clone(), bank.p.clone()
    k_eps, k_res = bank._key(1), bank._key(2)
    got = bank.predict(obs)
    for a in range(3):
        eps = hash_normal(k_eps[a].view(6, 1, 1) * 1048583 + bank._pu.view(1, 40, 24))
        ru = hash_uniform(k_res[a].view(6, 1) * 1048583 + bank._pu[:, 0].view(1, 40))
        with torch.no_grad():
            want, _ = bank.cells[a](obs[:, a, :3].contiguous(), (h0[a], p0[a]), eps, resample_u=ru)
        assert torch.allclose(got[:, a], want, rtol=1e-5, atol=1e-6), a
    # new weights are picked up (the stacked copies are rebuilt)
    sd = {k: v * 1.5 for k, v in bank.state_dict(1).items()}
    bank.load_state_dict(1, sd)
    bank.reset()
    assert not torch.allclose(bank.predict(obs)[:, 1], got[:, 1])


def test_hash_normal_pairs_are_standard_and_uncorrelated():
    """hash_normal draws one hash per PAIR of units (Box-Muller's cosine and sine): both members are N(0, 1) and uncorrelated."""
    keys = (torch.arange(20000, dtype=torch.int64).view(-1, 1) * 1048583 + 17) * 4096 + torch.arange(24, dtype=torch.int64).view(1, -1)
    z = hash_normal(keys).double()
    assert z.shape == (20000, 24)
    assert abs(float(z.mean())) < 5e-3 and abs(float(z.var()) - 1.0) < 1e-2
    even, odd = z[:, 0::2].reshape(-1), z[:, 1::2].reshape(-1)
    assert abs(float((even * odd).mean())) < 5e-3                                   # cos / sin of one angle: uncorrelated
    assert abs(float((even ** 2 * odd ** 2).mean()) - 1.0) < 2e-2                   # ... and independent in the second moments
    assert abs(float((z ** 4).mean()) - 3.0) < 5e-2                                 # kurtosis of a normal
    assert float(z.abs().max()) < 6.0                                               # 24-bit u1: |z| <= sqrt(2 ln 2^24) = 5.77
