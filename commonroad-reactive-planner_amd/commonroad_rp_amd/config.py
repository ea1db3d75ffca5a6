"""Configuration dataclasses with the fields the hot path reads (reference:
commonroad_rp/utility/config.py:107-254; defaults identical).  The OmegaConf/YAML loader of the
reference is out of scope; ``from_dict`` accepts the parsed YAML mapping."""
from __future__ import annotations

import dataclasses
from typing import Any, List, Optional

import numpy as np


@dataclasses.dataclass
class PlanningConfiguration:
    dt: float = 0.1
    time_steps_computation: int = 60
    replanning_frequency: int = 3
    continuous_collision_check: bool = False
    factor: int = 1
    low_vel_mode_threshold: float = 4.0
    constraints_to_check: List[str] = dataclasses.field(
        default_factory=lambda: ["velocity", "acceleration", "kappa", "kappa_dot", "yaw_rate"])
    standstill_lookahead: int = 10


@dataclasses.dataclass
class SamplingConfiguration:
    sampling_method: int = 1
    longitudinal_mode: str = "velocity_keeping"
    num_sampling_levels: int = 4
    t_min: float = 0.4
    v_min: float = 0
    v_max: float = 0
    s_min: float = -1
    s_max: float = 1
    d_min: float = -3
    d_max: float = 3


@dataclasses.dataclass
class DebugConfiguration:
    save_plots: bool = False
    show_plots: bool = False
    draw_traj_set: bool = False
    logging_level: str = "INFO"
    multiproc: bool = False      # the GPU batch replaces the fork fan-out (reactive_planner.py:1084-1111)
    num_workers: int = 1


@dataclasses.dataclass
class VehicleConfiguration:
    """Vehicle 2 (BMW 320i) of commonroad-vehicle-models 3.0.2, the vehicle of every shipped config."""
    id_type_vehicle: int = 2
    length: float = 4.508
    width: float = 1.610
    wb_front_axle: float = 1.1562
    wb_rear_axle: float = 1.4227
    a_max: float = 11.5
    v_switch: float = 7.319
    delta_min: float = -1.066
    delta_max: float = 1.066
    v_delta_min: float = -0.4
    v_delta_max: float = 0.4
    wheelbase: float = 1.1562 + 1.4227

    def __post_init__(self):
        self.kappa_max = np.tan(self.delta_max) / self.wheelbase


@dataclasses.dataclass
class ReactivePlannerConfiguration:
    vehicle: VehicleConfiguration = dataclasses.field(default_factory=VehicleConfiguration)
    planning: PlanningConfiguration = dataclasses.field(default_factory=PlanningConfiguration)
    sampling: SamplingConfiguration = dataclasses.field(default_factory=SamplingConfiguration)
    debug: DebugConfiguration = dataclasses.field(default_factory=DebugConfiguration)
    scenario: Any = None
    planning_problem: Any = None

    @classmethod
    def from_dict(cls, d: Optional[dict]) -> "ReactivePlannerConfiguration":
        cfg = cls()
        for section, values in (d or {}).items():
            sub = getattr(cfg, section, None)
            if sub is None or not dataclasses.is_dataclass(sub):
                continue
            for k, v in (values or {}).items():
                if hasattr(sub, k):
                    setattr(sub, k, v)
        cfg.vehicle.__post_init__()
        return cfg
