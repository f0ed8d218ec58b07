"""Closest-distance queries against PyBullet bodies (reference utils/collision_detector.py:7-98): the minimum
contact distance between one link of a body and each obstacle / each other link, saturating at `max_distance`
when PyBullet reports no pair within range. Thin host glue; imported only when PyBullet is present."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence

import numpy as np
import pybullet as p

_CONTACT_DISTANCE = 8   # field of a getClosestPoints() record holding the signed distance


@dataclass
class CollisionObject:
    """A (body uid, link index) pair distances are measured from."""
    body: int
    link: int


def _closest(body_a, body_b, max_distance: float, **links) -> float:
    points = p.getClosestPoints(body_a, body_b, distance=max_distance, **links)
    return max_distance if len(points) == 0 else float(np.min([pt[_CONTACT_DISTANCE] for pt in points]))


class CollisionDetector:

    def __init__(self, collision_object: CollisionObject, obstacle_ids: Sequence[int]):
        self.collision_object = collision_object
        self.obstacles = list(obstacle_ids)

    def compute_distances(self, max_distance: float = 10.0) -> np.ndarray:
        """One distance per obstacle body (reference :33-61)."""
        co = self.collision_object
        return np.array([_closest(co.body, obstacle, max_distance, linkIndexA=co.link) for obstacle in self.obstacles])

    def compute_collisions_in_manipulator(self, affected_joints: List[int], max_distance: float = 10.) -> np.ndarray:
        """Distances from this link to the other links of the same body, skipping itself and its two neighbours,
        which are always in contact (reference :63-98)."""
        co = self.collision_object
        return np.array([_closest(co.body, co.body, max_distance, linkIndexA=co.link, linkIndexB=j)
                         for j in affected_joints if abs(j - co.link) > 1])
