"""Sampling plug-in surface (reference: commonroad_rp/sampling.py:28-118 1-D sample sets,
:121-175 SamplingSpace ABC, :178-270 FixedIntervalSampling, :400-408 factory).

What the GPU path adds is ``FixedIntervalSampling.grids_at_level``: the three 1-D grids of a level
*in the reference's iteration order* -- the order of Python sets of floats, which decides the list
index of every candidate and therefore every tie (hard part 1).  The sets are
built by the same expressions as in the reference, so the iteration order is the same by
construction.  ``generate_trajectories_at_level`` (the reference's object-per-candidate interface)
is kept for plug-ins and tools; the planner itself never calls it for ``FixedIntervalSampling``.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Dict, List, Tuple

import numpy as np

from .polynomial_trajectory import QuarticTrajectory, QuinticTrajectory
from .trajectories import TrajectorySample


class Sampling(ABC):
    def __init__(self, low: float, up: float, num_sampling_levels: int):
        assert np.greater_equal(up, low), f"<Sampling>: Upper sampling bound is not greater than lower bound! up = {up} , low = {low}"
        assert isinstance(num_sampling_levels, int) and num_sampling_levels > 0, "<Sampling: number of samples must be positive integer>"
        self.low, self.up = low, up
        self._n_samples = num_sampling_levels
        self._dict_level_to_sample_set: Dict[int, set] = dict()
        self._sample()

    @abstractmethod
    def _sample(self):
        ...

    def samples_at_level(self, sampling_level: int = 0) -> set:
        assert 0 <= sampling_level < self.num_sampling_levels, f"<Sampling>: Provided sampling level is incorrect! stage = {sampling_level}"
        return self._dict_level_to_sample_set[sampling_level]

    @property
    def num_sampling_levels(self) -> int:
        return self._n_samples


class _LinspaceSampling(Sampling):
    """3, 5, 9, 17, ... equidistant samples per level (sampling.py:80-84, 95-99)."""

    def _sample(self):
        n = 3
        for i in range(self.num_sampling_levels):
            self._dict_level_to_sample_set[i] = set(np.linspace(self.low, self.up, n))
            n = (n * 2) - 1


class VelocitySampling(_LinspaceSampling):
    pass


class PositionSampling(_LinspaceSampling):
    pass


class TimeSampling(Sampling):
    """Multiples of dt between t_min and the horizon, denser per level (sampling.py:102-118)."""

    def __init__(self, low: float, up: float, num_sampling_levels: int, dt: float):
        self.dT = dt
        assert low >= 2 * self.dT, "<TimeSampling: lower bound of time sampling must be greater-equal than the given time step>"
        super().__init__(low, up, num_sampling_levels)

    def _sample(self):
        for i in range(self.num_sampling_levels):
            step_size = int((1 / (i + 1)) / self.dT)
            samp = set(np.arange(self.low, round(self.up + self.dT, 2), step_size * self.dT))
            samp.discard(round(self.up + self.dT, 2))
            self._dict_level_to_sample_set[i] = samp


class SamplingSpace(ABC):
    """Plug-in interface: any subclass only has to provide ``generate_trajectories_at_level``."""

    def __init__(self, num_sampling_levels: int):
        self._num_sampling_levels = num_sampling_levels
        self.samples_t = None
        self.samples_d = None
        self.samples_v = None
        self.samples_s = None

    @property
    def num_sampling_levels(self) -> int:
        return self._num_sampling_levels

    @abstractmethod
    def generate_trajectories_at_level(self, level_sampling: int, x_0_lon: np.ndarray, x_0_lat: np.ndarray,
                                       longitudinal_mode: str, low_vel_mode: bool) -> List[TrajectorySample]:
        ...


class FixedIntervalSampling(SamplingSpace):
    def __init__(self, config):
        super().__init__(config.sampling.num_sampling_levels)
        cs = config.sampling
        self.dt = config.planning.dt
        self.horizon = config.planning.dt * config.planning.time_steps_computation
        self._longitudinal_mode = None
        self.samples_t = TimeSampling(cs.t_min, self.horizon, cs.num_sampling_levels, self.dt)
        self.samples_d = PositionSampling(cs.d_min, cs.d_max, cs.num_sampling_levels)
        self.samples_v = VelocitySampling(cs.v_min, cs.v_max, cs.num_sampling_levels)
        self.samples_s = PositionSampling(cs.s_min, cs.s_max, cs.num_sampling_levels)

    def _get_lon_samples(self, level_sampling: int, longitudinal_mode: str) -> set:
        if longitudinal_mode == "velocity_keeping":
            return self.samples_v.samples_at_level(level_sampling)
        if longitudinal_mode == "stopping":
            return self.samples_s.samples_at_level(level_sampling)
        raise AttributeError(f"<FixedIntervalSampling>: specified longitudinal mode {longitudinal_mode} is invalid.")

    # ---- batch view for the GPU path ----------------------------------------------------------------
    def grids_at_level(self, level_sampling: int, x_0_lon, x_0_lat, longitudinal_mode: str) \
            -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
        """(T, traj_len, L, D) in the iteration order of the reference's triple loop
        (sampling.py:218-226).  Candidate index = (iT * len(L) + iL) * len(D) + iD.
        ``traj_len`` follows reactive_planner.py:733,748.  In stopping mode the longitudinal samples
        not ahead of the vehicle are dropped, which is what ``filter_goals_behind`` does to the list
        (trajectories.py:545-550, reactive_planner.py:1076-1077)."""
        T, traj_len = self._cached_T(self.samples_t.samples_at_level(level_sampling))
        lon_set = self._get_lon_samples(level_sampling, longitudinal_mode)
        L = self._cached_array(lon_set)
        if longitudinal_mode == "stopping":
            L = L[L > x_0_lon[0]]
        d_set = self.samples_d.samples_at_level(level_sampling)
        d0 = x_0_lat[0]
        if d0 in d_set:
            D = self._cached_array(d_set)
        else:   # set.union({d0}) keeps the iteration order of the copy and appends / hashes d0 in
            u = d_set.union({d0})
            D = np.fromiter(u, dtype=np.float64, count=len(u))
        return T, traj_len, L, D

    # The sample sets only change when a set_*_sampling_parameters call replaces them, so their array form
    # (in set-iteration order) is cached per set object; the set is kept alive next to it, hence ids are unique.
    def _cached_array(self, sample_set: set) -> np.ndarray:
        cache = self.__dict__.setdefault("_set_cache", {})
        hit = cache.get(id(sample_set))
        if hit is None or hit[0] is not sample_set or len(hit[1]) != len(sample_set):
            if len(cache) > 64:
                cache.clear()
            hit = cache[id(sample_set)] = (sample_set, np.array([float(v) for v in sample_set], dtype=np.float64))
        return hit[1]

    def _cached_T(self, t_set: set):
        cache = self.__dict__.setdefault("_t_cache", {})
        hit = cache.get(id(t_set))
        if hit is None or hit[0] is not t_set or len(hit[1]) != len(t_set):
            if len(cache) > 64:
                cache.clear()
            T = np.array([float(t) for t in t_set], dtype=np.float64)
            traj_len = np.array([len(np.arange(0, np.round(t + self.dt, 5), self.dt)) for t in T], dtype=np.int32)
            hit = cache[id(t_set)] = (t_set, T, traj_len)
        return hit[1], hit[2]

    # ---- reference interface (one object per candidate) --------------------------------------------
    def generate_trajectories_at_level(self, level_sampling: int, x_0_lon, x_0_lat, longitudinal_mode: str,
                                       low_vel_mode: bool) -> List[TrajectorySample]:
        self._longitudinal_mode = longitudinal_mode
        out = []
        x_0_lon, x_0_lat = np.array(x_0_lon, dtype=float), np.array(x_0_lat, dtype=float)
        for t in self.samples_t.samples_at_level(level_sampling):
            for lon_sample in self._get_lon_samples(level_sampling, longitudinal_mode):
                if longitudinal_mode == "velocity_keeping":
                    lon = QuarticTrajectory(tau_0=0, delta_tau=t, x_0=x_0_lon, x_d=np.array([lon_sample, 0.0]))
                else:
                    lon = QuinticTrajectory(tau_0=0, delta_tau=t, x_0=x_0_lon, x_d=np.array([lon_sample, 0.0, 0.0]))
                for d in self.samples_d.samples_at_level(level_sampling).union({x_0_lat[0]}):
                    tau = t
                    if low_vel_mode:
                        s_lon_goal = lon.evaluate_state_at_tau(t)[0] - x_0_lon[0]
                        tau = t if s_lon_goal <= 0 else s_lon_goal
                    lat = QuinticTrajectory(tau_0=0, delta_tau=tau, x_0=x_0_lat, x_d=np.array([d, 0.0, 0.0]))
                    out.append(TrajectorySample(self.horizon, self.dt, lon, lat))
        return out


class CorridorSampling(SamplingSpace):
    """Adaptive corridor sampling needs the optional CommonRoad-Reach package (sampling.py:273-397);
    out of scope here -- same failure mode as the reference without it."""

    def __init__(self, config):
        raise ImportError("<CorridorSampling>: Please install CommonRoad-Reach to use adaptive corridor sampling!")

    def generate_trajectories_at_level(self, *a, **k):
        raise NotImplementedError


def sampling_space_factory(config):
    method = config.sampling.sampling_method
    if method == 1:
        return FixedIntervalSampling(config)
    if method == 2:
        return CorridorSampling(config)
    raise ValueError("Invalid sampling method specified")
