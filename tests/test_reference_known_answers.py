"""Known answers of the reference's OWN unit tests for the heat-map side (unit_tests/test_RADTEAM_core.py), as data:
IntensityEstimator medians (:56-76), Normalizer.normalize_incremental_logscale (:287-299), MapsBuffer visit counts (:536-563),
obstacle map (:565-577), location maps (:451-497).  The readings-map expectations of that file (:499-527) describe an older
min-max normalisation than the code it ships with (RADTEAM_core.py:844-872 standardises with Welford) and are not used."""
import math

import numpy as np
import pytest

from oracle.maps_oracle import MapsOracle, logscale, median


def test_intensity_estimator_medians():
    assert median([1000.0, 2000.0]) == 1500
    assert median([1000.0, 2000.0, 500.0]) == 1000
    assert median([1000.0, 2000.0, 300.0, 300.0]) == 650           # get_min() after the fourth update (:93-97)


def test_normalizer_logscale_known_answers():
    assert logscale(4.0, 10, 2) == pytest.approx(0.598104004)
    assert logscale(18.0, 10, 2) == 1
    assert logscale(1.0, 10, 2) == pytest.approx(0.366725791)


def _obs(reading, cx, cy, det=None):
    o = np.zeros(11)
    o[0] = reading
    o[1], o[2] = (cx + 0.5) / 22.0, (cy + 0.5) / 22.0              # a point inside cell (cx, cy) of the 22-per-unit grid
    if det is not None:
        o[3:] = det
    return o


def test_visit_count_map_known_answers():
    """MapsBuffer(steps_per_episode=120, number_of_agents=2): base = 242; one agent visiting (0,1), then (0,2) twice, then
    (0,3) for 2 * 120 visits."""
    m = MapsOracle(steps_per_episode=120, number_of_agents=2)
    m.observation_to_map({0: _obs(10.0, 0, 1)}, 0, (0.0, 0.0))
    assert m.visits[0][1] == pytest.approx(0.11212191) and m.shadow[(0, 1)] == 2
    m.observation_to_map({0: _obs(10.0, 0, 2)}, 0, (0.0, 0.0))
    assert m.visits[0][1] == pytest.approx(0.11212191) and m.visits[0][2] == pytest.approx(0.11212191) and m.shadow[(0, 2)] == 2
    m.observation_to_map({0: _obs(10.0, 0, 2)}, 0, (0.0, 0.0))
    assert m.visits[0][2] == pytest.approx(0.22424382) and m.shadow[(0, 2)] == 4
    for _ in range(2 * 120):
        m.observation_to_map({0: _obs(10.0, 0, 3)}, 0, (0.0, 0.0))
    assert m.visits[0][3] == pytest.approx(0.99865758) and m.shadow[(0, 3)] == 2 * 2 * 120


def test_obstacle_and_location_maps_known_answers():
    det = [0.1, 0.1, 0.1, 0.1, 0.05, 0.1, 0.1, 0.1]
    m = MapsOracle(steps_per_episode=120, number_of_agents=2)
    m.observation_to_map({0: _obs(1500.0, 0, 1, det)}, 0, (0.0, 0.0))
    assert m.obstacles[0][1] == pytest.approx(0.1)                  # the last non-zero detection wins (:910-932)
    assert m.location[0][1] == 1.0 and np.delete(m.location.ravel(), 1).max() == 0.0
    assert m.combined[0][1] == 1.0 and np.delete(m.combined.ravel(), 1).max() == 0.0
    m.observation_to_map({0: _obs(1500.0, 0, 2, det)}, 0, (0.0, 0.0))
    assert m.obstacles[0][1] == pytest.approx(0.1) and m.obstacles[0][2] == pytest.approx(0.1)
    assert m.location[0][2] == 1.0 and m.location[0][1] == 0.0 and m.combined[0][2] == 1.0 and m.combined[0][1] == 0.0
    # seen from another owner the same moves land on the `others` map
    o = MapsOracle(steps_per_episode=120, number_of_agents=2)
    o.observation_to_map({0: _obs(1.0, 0, 1)}, 1, (0.0, 0.0))
    o.observation_to_map({0: _obs(1.0, 0, 2)}, 1, (0.0, 0.0))
    assert o.others[0][2] == 1.0 and np.delete(np.delete(o.others.ravel(), 2), 1).max() == 0.0 and o.others[0][1] == 0.0
