"""Config — the option fields of the reference's `Config` value class that parameterise the hot path.

Mirrors config/Config.m:3-51 (defaults) and the derived flag `are_any_obstacles_non_convex`
(config/Config.m:71-87), which selects the constraint checker in
OptimizerInterface.set_constraint_checker (hlc/optimizer/OptimizerInterface.m:36-46).
"""
from dataclasses import dataclass, field
from enum import Enum


class ScenarioType(Enum):  # config/enums/ScenarioType.m
    commonroad = "commonroad"
    circle = "circle"


class MpaType(Enum):  # config/enums/MpaType.m
    single_speed = "single_speed"
    triple_speed = "triple_speed"
    realistic = "realistic"


class OptimizerType(Enum):  # config/enums/OptimizerType.m:3-6 plus this backend's member
    MatlabOptimal = "MatlabOptimal"
    MatlabSampled = "MatlabSampled"
    HipOptimal = "HipOptimal"
    HipSampled = "HipSampled"


class ConstraintFromSuccessor(Enum):  # config/enums/ConstraintFromSuccessor.m
    none = "none"
    area_of_standstill = "area_of_standstill"
    area_of_previous_trajectory = "area_of_previous_trajectory"


@dataclass
class Config:
    scenario_type: ScenarioType = ScenarioType.commonroad  # Config.m:6
    amount: int = 20  # Config.m:8
    T_end: float = 20.0  # Config.m:9
    path_ids: list = field(default_factory=list)  # Config.m:10
    is_prioritized: bool = True  # Config.m:22
    max_num_CLs: int = 99  # Config.m:28
    optimizer_type: OptimizerType = OptimizerType.HipOptimal  # Config.m:30 (reference default MatlabOptimal)
    dt_seconds: float = 0.2  # Config.m:32
    Hp: int = 6  # Config.m:33
    mpa_type: MpaType = MpaType.single_speed  # Config.m:35
    constraint_from_successor: ConstraintFromSuccessor = ConstraintFromSuccessor.area_of_standstill  # Config.m:37
    recursive_feasibility: bool = True  # Config.m:47
    time_per_tick: float = 0.01  # Config.m:48
    offset: float = 0.01  # Config.m:49
    # backend sizing (no reference counterpart)
    device: int = 0
    max_nodes: int = 0
    max_vehicles: int = 0
    trace_pops: int = 0

    @property
    def tick_per_step(self) -> int:  # Config.m:63-65
        return int(round(self.dt_seconds / self.time_per_tick))

    @property
    def k_end(self) -> int:  # Config.m:67-69
        import math

        return int(math.floor(self.T_end / self.dt_seconds))

    @property
    def are_any_obstacles_non_convex(self) -> bool:  # Config.m:71-87
        if self.scenario_type == ScenarioType.circle or not self.is_prioritized:
            return False
        return True
