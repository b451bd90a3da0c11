"""World objects of `Drone.step(..., object_list=[...])`, with the reference's constructor signatures.

Analytic counterparts of the reference's world classes, keeping only what `handle_collisions`
(/root/reference/src/utils/components.py:198-214) evaluates - a signed distance and a normal:

    Ground(size, resolution, random=False)                                plane z = 0        components.py:646-680
    Cylinder(position, radius, height, angle_resolution, height_resolution, random=False)    components.py:685-729
    Target(position, radius, nu, path=None)                               sphere; `update()` follows a circular
                                                                          path                components.py:753-778
    Gate(position, rotation_matrix, size, shape="rectangle", resolution=17)                  components.py:780-831
    Trail(trail_length=-1)                                                                   components.py:631-644

The rendering arguments (sizes of point clouds, icosphere subdivision, resolutions) are accepted so the
reference's own call sites (src/core/simulator.py:53-58, src/utils/generators.py) construct these classes
unchanged, and are ignored: point clouds, rendering and bounding boxes are out of scope.  Gates and the
Trail never collide in the reference (components.py:202); they exist here only so that an `object_list`
built by the reference's code can be passed as is - `to_rows` skips them.
"""
from __future__ import annotations

from typing import Any, Optional, Sequence, Tuple

import numpy as np

from . import _lib


class Ground:
    def __init__(self, size: float = 0.0, resolution: int = 0, random: bool = False):
        self.size, self.resolution = size, resolution             # point-cloud arguments: unused

    position = property(lambda self: np.zeros(3))                 # components.py:667-669

    def as_row(self) -> Tuple[float, ...]:
        return (_lib.OBJ_GROUND, 0.0, 0.0, 0.0, 0.0, 0.0)


class Cylinder:
    def __init__(self, position: Sequence[float], radius: float, height: float, angle_resolution: int = 0,
                 height_resolution: int = 0, random: bool = False):
        assert radius > 0, "radius must be positive"              # components.py:688-689
        assert height > 0, "height must be positive"
        self.position = np.asarray(position, dtype=np.float64)
        self.radius, self.height = float(radius), float(height)

    def as_row(self) -> Tuple[float, ...]:
        p = self.position
        return (_lib.OBJ_CYLINDER, float(p[0]), float(p[1]), float(p[2]), self.radius, self.height)


def circular_path(center: Sequence[float], radius: float, resolution: int) -> np.ndarray:
    """helper_functions.generate_circular_path (helper_functions.py:151-153)."""
    theta = np.linspace(0, 2 * np.pi, resolution + 1)[:-1]
    return np.vstack((np.cos(theta) * radius, np.sin(theta) * radius, np.zeros_like(theta))).T + np.array(center)


class Target:
    """Sphere target; with `path={"radius": r, "resolution": k}` it follows a circle around its
    initial position, one path point per `update()` (components.py:741-772).  `nu` (icosphere
    subdivision of the rendered mesh) is accepted and ignored."""

    def __init__(self, position: Sequence[float], radius: float, nu: Any = None, path: Optional[dict] = None):
        if isinstance(nu, dict) and path is None:                 # Target(position, radius, path_dict)
            nu, path = None, nu
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


class Gate:
    """Never collides (components.py:202): carried through an object_list and skipped."""
    collides = False

    def __init__(self, position, rotation_matrix, size, shape: str = "rectangle", resolution: int = 17):
        self.position, self.rotation_matrix, self.size, self.shape = position, rotation_matrix, size, shape


class Trail:
    """Never collides (components.py:202): carried through an object_list and skipped."""
    collides = False

    def __init__(self, trail_length: int = -1):
        self.trail_length = trail_length


def to_rows(object_list) -> Tuple[Tuple[float, ...], ...]:
    """object_list -> (type, x, y, z, radius, height) rows in list order; Gate / Trail entries (ours, or
    any object whose class is named so, e.g. the reference's own) are skipped like components.py:202 does."""
    rows = []
    for o in object_list:
        if getattr(o, "collides", True) is False or type(o).__name__ in ("Gate", "Trail"):
            continue
        rows.append(o.as_row() if hasattr(o, "as_row") else tuple(float(x) for x in o))
    return tuple(rows)
