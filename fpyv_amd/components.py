"""Same import shape as the reference's `utils.components` for the classes that are on (or feed)
the per-drone step: `from fpyv_amd.components import Drone, Ground, Cylinder, Target, Gate, Trail, PID`.

    reference class (src/utils/components.py)      here
    Drone   :72-248                                 fpyv_amd.env.DroneBatch  (N drones per object; `Drone(params)`
                                                    takes the same params dict, plus num_envs= / device=)
    PID     :15-54                                  fpyv_amd.pid.PID         (N controllers per object)
    Ground  :646-683, Cylinder :685-729,            fpyv_amd.objects         (reference constructor signatures;
    Target  :753-778, Gate :780-831, Trail :631-644                           distance + normal only, Gate/Trail inert)

`Racer` (/root/reference/tests/racer_drone_test.py:68-103) maps to fpyv_amd.env.RacerBatch.
Camera and the guidance methods are out of scope.
"""
from .env import DroneBatch as Drone, RacerBatch as Racer, FpvVecEnv  # noqa: F401
from .objects import Cylinder, Gate, Ground, Target, Trail  # noqa: F401
from .pid import PID  # noqa: F401

__all__ = ["Drone", "Racer", "FpvVecEnv", "Ground", "Cylinder", "Target", "Gate", "Trail", "PID"]
