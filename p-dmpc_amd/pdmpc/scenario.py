"""Synthetic scenarios that feed the optimizer (the reference builds these in scenarios/**).

circle_scenario      scenarios/free_space/Circle.m:7-44, literally.
Road-network scenarios live in road_network.py (lab map fixture) — SURVEY.md 8(d) C2..C5.
"""
import math
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from .config import Config
from .mpa import VEHICLE_LENGTH, VEHICLE_WIDTH, get_mpa


@dataclass
class Vehicle:  # scenarios/Vehicle.m:4-19
    x_start: float = 0.0
    y_start: float = 0.0
    yaw_start: float = 0.0
    reference_path: np.ndarray = field(default_factory=lambda: np.zeros((0, 2)))
    reference_speed: float = 0.0
    Length: float = VEHICLE_LENGTH
    Width: float = VEHICLE_WIDTH
    lanelets_index: Optional[List[int]] = None
    points_index: Optional[np.ndarray] = None
    is_loop: bool = True


@dataclass
class Scenario:  # scenarios/Scenario.m (fields the controller reads)
    vehicles: List[Vehicle]
    obstacles: list = field(default_factory=list)  # Scenario.m:6-7
    dynamic_obstacle_area: list = field(default_factory=list)
    lanelet_boundary: Optional[list] = None  # per lanelet (left (P,2), right (P,2))
    adjacency_lanelets: Optional[np.ndarray] = None


def circle_scenario(options: Config) -> Scenario:
    """Circle.m:7-44: nVeh vehicles on a radius-2 circle around (2.25, 2) heading to the centre."""
    nVeh = options.amount
    radius = 2
    reference_speed = max(get_mpa(options).get_straight_speeds_of_mpa())  # :19-20
    vehicles = []
    for i in range(nVeh):
        yaw = math.pi * 2 / nVeh * i  # :17
        s, c = math.sin(yaw), math.cos(yaw)
        x_start = -c * radius + 2.25  # :27,31-33
        y_start = -s * radius + 2
        x_end = x_start + c * 2 * radius  # :35-36
        y_end = y_start + s * 2 * radius
        vehicles.append(
            Vehicle(
                x_start=x_start,
                y_start=y_start,
                yaw_start=yaw,
                reference_path=np.array([[x_start, y_start], [x_end, y_end]]),
                reference_speed=reference_speed,
            )
        )
    return Scenario(vehicles=vehicles)
