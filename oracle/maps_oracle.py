"""CPU ORACLE (test infrastructure, NOT product code): the RAD-TEAM heat-map builder.

Plain Python/numpy restatement of algos/multiagent/NeuralNetworkCores/RADTEAM_core.py (paths relative to
/root/reference):
    calculate_map_dimensions / calculate_resolution_accuracy   :61-76
    IntensityEstimator (median of the readings per cell)       :102-185
    StatisticStandardization (Welford + running max/min)       :188-277
    Normalizer.normalize_incremental_logscale                  :321-362
    MapsBuffer.observation_to_map / reset / _update_*          :532-616, :618-690, :692-932
Only tests/ may import it.

PINNED: tests/test_maps_oracle.py replays tests/golden/maps.npz, captured from the reference's own MapsBuffer
class (tests/golden/make_golden.py gen_maps), float32-exact on all seven maps for every step and owner, plus the
known answers of unit_tests/test_RADTEAM_core.py (inflate/deflate :421-449, Welford :144-236).

The location-prediction input (PFGRU) is an argument here exactly as in the reference.
"""
import math
from statistics import median

import numpy as np


def calculate_resolution_accuracy(resolution_multiplier, scale):      # RADTEAM_core.py:70-71
    return resolution_multiplier * 1 / scale


def calculate_map_dimensions(grid_bounds, resolution_accuracy, offset):   # :61-67
    return (int(grid_bounds[0] * resolution_accuracy) + int(offset * resolution_accuracy),
            int(grid_bounds[1] * resolution_accuracy) + int(offset * resolution_accuracy))


def logscale(current_value, base, increment_value=2):                  # Normalizer.normalize_incremental_logscale :321-362
    return (math.log(increment_value + current_value, base)) * 1 / math.log(increment_value * base, base)


class MapsOracle:
    """One MapsBuffer (one owner agent `id` decides location vs others maps at call time, as in the reference)."""

    def __init__(self, steps_per_episode, number_of_agents, resolution_accuracy=22.0, offset=0.22727272727272727,
                 grid_bounds=(1, 1)):
        self.ra = resolution_accuracy
        self.A = number_of_agents
        self.base = (steps_per_episode + 1) * number_of_agents            # :498
        self.dims = calculate_map_dimensions(grid_bounds, resolution_accuracy, offset)
        self._zero_maps()
        self.reset()

    def _zero_maps(self):
        z = lambda: np.zeros(self.dims, dtype=np.float32)
        self.prediction, self.combined, self.location, self.others = z(), z(), z(), z()
        self.readings_map, self.obstacles, self.visits = z(), z(), z()

    def reset(self):                                                      # :510-523 + ConversionTools.reset :385-392
        self._zero_maps()
        self.shadow = {}
        self.last_coords = {}
        self.last_prediction = ()
        self.cell_readings = {}
        # StatisticStandardization
        self.count, self.mean, self.sq, self.std = 0, 0.0, 0.0, 1.0

    def _inflate(self, o):                                                # :692-715
        if isinstance(o, np.ndarray):
            return (int(o[1] * self.ra), int(o[2] * self.ra))
        return (int(o[0] * self.ra), int(o[1] * self.ra))

    def _standardize_update(self, x):                                     # StatisticStandardization.update/standardize
        self.count += 1
        if self.count == 1:
            self.mean = x
        else:
            mean_new = self.mean + (x - self.mean) / self.count
            self.sq = self.sq + (x - self.mean) * (x - mean_new)
            self.mean = mean_new
            self.std = max(math.sqrt(self.sq / (self.count - 1)), 1)
        return (x - self.mean) / self.std

    def observation_to_map(self, observation, id, loc_prediction):       # :532-616
        for obs in observation.values():
            key = self._inflate(obs)
            self.cell_readings.setdefault(key, []).append(float(obs[0]))
        for agent_id in observation:
            cur = self._inflate(observation[agent_id])
            pred = self._inflate(loc_prediction)
            last = self.last_coords.get(agent_id)
            # prediction map (:747-766)
            if len(self.last_prediction) > 0:
                self.prediction[self.last_prediction[0]][self.last_prediction[1]] -= 1
            self.prediction[pred[0]][pred[1]] = 1
            # location maps (:768-842)
            if id == agent_id:
                if last:
                    self.location[last[0]][last[1]] -= 1
                self.location[cur[0]][cur[1]] = 1
            else:
                if last:
                    self.others[last[0]][last[1]] -= 1
                self.others[cur[0]][cur[1]] += 1
            if last:
                self.combined[last[0]][last[1]] -= 1
            self.combined[cur[0]][cur[1]] += 1
            # readings map (:844-872): median estimate -> Welford update -> z-score
            est = median(self.cell_readings[cur])
            self.readings_map[cur[0]][cur[1]] = self._standardize_update(est)
            # visit counts (:874-908)
            if cur in self.shadow:
                current = self.shadow[cur]
                self.shadow[cur] += 2
            else:
                current = 0
                self.shadow[cur] = 2
            self.visits[cur[0]][cur[1]] = logscale(current, self.base, 2)
            # obstacles (:910-932)
            det = observation[agent_id][3:]
            if np.count_nonzero(det) > 0:
                for d in det:
                    if d != 0:
                        self.obstacles[cur[0]][cur[1]] = d
            self.last_coords[agent_id] = cur
            self.last_prediction = pred
        return (self.prediction, self.location, self.others, self.readings_map, self.visits, self.obstacles, self.combined)
