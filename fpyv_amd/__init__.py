"""fpyv_amd - MI355X-native batched FPV drone physics stepper.

One fused HIP kernel (gfx950) advances N independent drones per call, behind a ctypes C ABI
(include/fpv_abi.h).  The host API mirrors the reference's `Drone.reset/step`
(/root/reference/src/utils/components.py:150,:220) in batched form, plus the gym-style
`reset()/step(action) -> (obs, reward, done, info)` surface used by the reference's env scripts.
"""
from .params import DroneParams, load_params, read_motor_test_report, MODE_DRONE, MODE_RACER  # noqa: F401

__all__ = ["DroneParams", "load_params", "read_motor_test_report", "MODE_DRONE", "MODE_RACER"]
__version__ = "0.3.0"
