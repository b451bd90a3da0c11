"""Collision objects for `Drone.step(..., object_list=[...])`.

Analytic counterparts of the reference's world objects, keeping only what `handle_collisions`
(/root/reference/src/utils/components.py:198-214) evaluates - a signed distance and a normal:

    Ground    plane z = 0                              components.py:646-680
    Cylinder  vertical, base at `position`             components.py:685-729
    Target    sphere; move it by assigning `.position` components.py:753-778 (path: helper_functions.py:151-153)

Point clouds, rendering and bounding boxes of the reference classes are out of scope.  Gates and
the Trail never collide in the reference (components.py:202) and have no counterpart here.
"""
from __future__ import annotations

import dataclasses
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _lib


@dataclasses.dataclass
class Ground:
    def as_row(self) -> Tuple[float, ...]:
        return (_lib.OBJ_GROUND, 0.0, 0.0, 0.0, 0.0, 0.0)


@dataclasses.dataclass
class Cylinder:
    position: Sequence[float]
    radius: float
    height: float

    def __post_init__(self):
        assert self.radius > 0, "radius must be positive"      # components.py:688-689
        assert self.height > 0, "height must be positive"

    def as_row(self) -> Tuple[float, ...]:
        p = self.position
        return (_lib.OBJ_CYLINDER, float(p[0]), float(p[1]), float(p[2]), float(self.radius), float(self.height))


def circular_path(center: Sequence[float], radius: float, resolution: int) -> np.ndarray:
    """helper_functions.generate_circular_path (helper_functions.py:151-153)."""
    theta = np.linspace(0, 2 * np.pi, resolution + 1)[:-1]
    return np.vstack((np.cos(theta) * radius, np.sin(theta) * radius, np.zeros_like(theta))).T + np.array(center)


class Target:
    """Sphere target; with `path={"radius": r, "resolution": k}` it follows a circle around its
    initial position, one path point per `update()` (components.py:741-772)."""

    def __init__(self, position: Sequence[float], radius: float, path: Optional[dict] = None):
        self.position = np.asarray(position, dtype=np.float64)
        self.radius = float(radius)
        self._path = circular_path(self.position, **path) if path is not None else None
        self._count = 0

    def update(self) -> None:
        if self._path is not None:
            self.position = self._path[self._count % len(self._path)]
            self._count += 1

    def as_row(self) -> Tuple[float, ...]:
        p = self.position
        return (_lib.OBJ_SPHERE, float(p[0]), float(p[1]), float(p[2]), self.radius, 0.0)


def to_rows(object_list) -> Tuple[Tuple[float, ...], ...]:
    return tuple(o.as_row() if hasattr(o, "as_row") else tuple(float(x) for x in o) for o in object_list)
