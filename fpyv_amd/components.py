"""Same import shape as the reference's `utils.components` for the classes that are on (or feed)
the per-drone step: `from fpyv_amd.components import Drone, Ground, Cylinder, Target`.

    reference class (src/utils/components.py)      here
    Drone   :72-248                                 fpyv_amd.env.DroneBatch  (N drones per object)
    Ground  :646-683, Cylinder :685-729,            fpyv_amd.objects         (distance + normal only)
    Target  :753-778

`Racer` (/root/reference/tests/racer_drone_test.py:68-103) maps to fpyv_amd.env.RacerBatch.
Camera, Trail, Gate, PID and the guidance methods are out of scope.
"""
from .env import DroneBatch as Drone, RacerBatch as Racer, FpvVecEnv  # noqa: F401
from .objects import Cylinder, Ground, Target  # noqa: F401

__all__ = ["Drone", "Racer", "FpvVecEnv", "Ground", "Cylinder", "Target"]
