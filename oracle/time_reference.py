#!/usr/bin/env python3
"""Time the REFERENCE's own Python Drone.step in the build container (it cannot travel to the GPU
box): 1 drone, dt = 1 ms, zero sticks, object_list = [], stdout redirected.  TEST INFRASTRUCTURE.

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python3 -W ignore oracle/time_reference.py
"""
import contextlib
import copy
import io
import json
import os
import platform
import sys
import time

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import import_reference, ref_params  # noqa: E402

yaml_helper, Drone, _, racer_mod = import_reference()
P = ref_params(yaml_helper, 1000)
steps = 10000
with contextlib.redirect_stdout(io.StringIO()):
    d = Drone(copy.deepcopy(P))
    d.reset(position=np.array([0, 0, 10.0]), velocity=np.array([1.0, 0, 0]), ypr=np.zeros(3))
    a, w = np.zeros(4), np.zeros(3)
    t0 = time.perf_counter()
    for _ in range(steps):
        d.step(action=a, wind_velocity_vector=w, object_list=[])
    dt_drone = time.perf_counter() - t0
    env = racer_mod.Racer(prop_size_inch=5, pid_values={"roll": [2, 0, 0], "pitch": [2, 0, 0], "yaw": [0.1, 0, 0]})
    env.reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        env.step(action=[80, 10, 0, 0])
    dt_racer = time.perf_counter() - t0
cpu = [ln.split(":")[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
out = {"drone_step_env_steps_per_s": steps / dt_drone, "racer_step_env_steps_per_s": steps / dt_racer, "steps": steps,
       "cores_used": 1, "cpu": cpu, "python": platform.python_version(), "numpy": np.__version__,
       "end_state_position": [float(x) for x in d.state[:3]],
       "note": "reference /root/reference/src/utils/components.py:220-248 (Drone.step) and "
               "tests/racer_drone_test.py:95-103 (Racer.step), float64 NumPy, single thread"}
print(json.dumps(out, indent=1))
