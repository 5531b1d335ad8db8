"""The heat-map oracle (oracle/maps_oracle.py) against golden vectors captured from the reference's MapsBuffer
(tests/golden/maps.npz) and the reference's own unit-test known answers."""
import os

import numpy as np

from oracle.maps_oracle import MapsOracle, calculate_map_dimensions, calculate_resolution_accuracy, logscale


def test_constants_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "maps.npz"))
    ra = calculate_resolution_accuracy(0.01, 1 / 2200.0)
    assert ra == float(g["ra"]) == 22.0
    off = (1 / 2200.0) * 500.0
    assert off == float(g["offset"])
    assert calculate_map_dimensions((1, 1), ra, off) == tuple(int(v) for v in g["map_dim"]) == (27, 27)
    base = int(g["base"])
    for i, c in enumerate(range(0, 2 * base, 2)):
        assert logscale(c, base, 2) == g["logscale"][i]


def test_inflate_known_answers():
    """unit_tests/test_RADTEAM_core.py:421-449 style: int(coordinate * resolution_accuracy)."""
    m = MapsOracle(steps_per_episode=120, number_of_agents=2)
    o = np.array([1500.0, 0.5, 0.25, 0, 0, 0, 0, 0, 0, 0, 0])
    assert m._inflate(o) == (11, 5) and m._inflate((0.5, 0.25)) == (11, 5)
    assert m.dims == (27, 27) and m.base == 242


def test_maps_match_reference_step_by_step(golden_dir):
    g = dict(np.load(os.path.join(golden_dir, "maps.npz")).items())
    A, L = int(g["A"]), int(g["L"])
    bufs = [MapsOracle(steps_per_episode=L, number_of_agents=A, resolution_accuracy=float(g["ra"]), offset=float(g["offset"]))
            for _ in range(A)]
    for t in range(g["obs"].shape[0]):
        od = {i: g["obs"][t, i] for i in range(A)}
        pred = (float(g["pred"][t, 0]), float(g["pred"][t, 1]))
        for i in range(A):
            maps = bufs[i].observation_to_map(od, i, pred)
            for k in range(7):
                assert np.array_equal(maps[k], g["maps"][t, i, k]), (t, i, k)
        if g["reset_after"][t]:
            for b in bufs:
                b.reset()


def test_tall_linear_two_step_weight_gradient_matches_autograd():
    """maps._LinearTall (the Linear layers behind the CNN trunk at update-chunk sizes): same outputs and gradients as nn.Linear."""
    import torch
    from radiation_ppo_amd.maps import _LinearTall, _head
    torch.manual_seed(0)
    for fin, fout in ((32, 16), (16, 8), (48, 32)):
        lin = torch.nn.Linear(fin, fout).double()
        x = torch.randn(65536, fin, dtype=torch.float64, requires_grad=True)
        g = torch.randn(65536, fout, dtype=torch.float64)
        y_ref = lin(x)
        gx_ref, gw_ref, gb_ref = torch.autograd.grad(y_ref, (x, lin.weight, lin.bias), g)
        y = _LinearTall.apply(x, lin.weight, lin.bias)
        gx, gw, gb = torch.autograd.grad(y, (x, lin.weight, lin.bias), g)
        assert torch.equal(y, y_ref) and torch.allclose(gx, gx_ref, rtol=1e-12, atol=1e-12)
        assert torch.allclose(gw, gw_ref, rtol=1e-10, atol=1e-10) and torch.allclose(gb, gb_ref, rtol=1e-10, atol=1e-10)
        assert _head(lin, x).grad_fn is not None and "LinearTall" in type(_head(lin, x).grad_fn).__name__
    small = torch.randn(100, 32, dtype=torch.float64, requires_grad=True)
    assert "LinearTall" not in type(_head(torch.nn.Linear(32, 16).double(), small).grad_fn).__name__      # small batches: plain nn.Linear
