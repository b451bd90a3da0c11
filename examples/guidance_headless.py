#!/usr/bin/env python3
"""The guidance branch of the reference's simulator loop (/root/reference/src/core/simulator.py:104-110) for N
drones at once, headless:

    rot_mat, force_size = <guidance law>(drone, target)
    drone.step(action=action, wind_velocity_vector=wind, object_list=object_list,
               rotation_matrix=rot_mat, thrust_force=force_size)                      # components.py:230-232

The reference derives (rot_mat, force_size) from the camera image of the target
(Drone.calculate_needed_force_orientation, out of scope: rendering); here a plain geometric law stands in for it -
point the body z axis along "gravity compensation + a pull towards the target", thrust = that vector's length - so
that the example exercises the same call with the same argument shapes: rotation_matrix [N,3,3], thrust_force [N].
Drones further than --engage metres from the target are left to their sticks (NaN thrust_force = not overridden),
like the released gamepad button of simulator.py:104.

Usage:  python examples/guidance_headless.py --drones 65536 --steps 3000
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fpyv_amd import load_params  # noqa: E402
from fpyv_amd.components import Drone, Ground, Target  # noqa: E402


def look_along(z_axis: torch.Tensor, heading: torch.Tensor) -> torch.Tensor:
    """[N,3,3] rotation matrices (body -> world, columns = body axes) whose third column is z_axis / |z_axis| and
    whose first column is the heading made orthogonal to it."""
    z = z_axis / z_axis.norm(dim=1, keepdim=True)
    x = heading - (heading * z).sum(dim=1, keepdim=True) * z
    x = x / x.norm(dim=1, keepdim=True).clamp_min(1e-6)
    y = torch.linalg.cross(z, x)
    return torch.stack([x, y, z], dim=2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--drones", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--fps", type=float, default=250.0)
    ap.add_argument("--engage", type=float, default=60.0, help="guidance takes over within this distance of the target [m]")
    ap.add_argument("--gain", type=float, default=1.5, help="pull towards the target [1/s^2]")
    ap.add_argument("--damp", type=float, default=2.0, help="velocity damping [1/s]")
    a = ap.parse_args()
    dev = "cuda:0"
    rng = np.random.default_rng(0)
    params = load_params(fps=a.fps)
    target = Target(np.array([0.0, 0.0, 12.0]), 1.0, 5, {"radius": 25.0, "resolution": 20000})
    ground = Ground(size=60, resolution=50, random=False)
    drone = Drone(params, num_envs=a.drones, device=dev)
    drone.reset(position=rng.uniform([-40, -40, 5], [40, 40, 30], (a.drones, 3)).astype(np.float32),
                velocity=np.zeros(3), ypr=np.zeros(3))
    sticks = torch.tensor([0.0, 0.0, 0.0, -0.646], device=dev).expand(a.drones, 4).contiguous()   # hover throttle when not engaged
    g = torch.tensor([0.0, 0.0, params.gravity], device=dev)
    heading = torch.tensor([1.0, 0.0, 0.0], device=dev).expand(a.drones, 3)
    wind = np.zeros(3)
    closest = torch.full((a.drones,), float("inf"), device=dev)

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        target.update()                                                     # simulator.py:87
        tpos = torch.as_tensor(np.asarray(target.position, dtype=np.float32), device=dev)
        to_target = tpos - drone.position
        dist = to_target.norm(dim=1)
        closest = torch.minimum(closest, dist)
        want = g + a.gain * to_target - a.damp * drone.velocity             # acceleration the thrust has to supply
        rot_mat = look_along(want, heading)
        force_size = params.mass * want.norm(dim=1)
        force_size = torch.where(dist < a.engage, force_size, torch.full_like(force_size, float("nan")))
        drone.step(action=sticks, wind_velocity_vector=wind, object_list=[target, ground],
                   rotation_matrix=rot_mat, thrust_force=force_size, return_imu=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{a.drones} drones x {a.steps} guided steps (dt = {params.dt * 1e3:.1f} ms) in {dt:.3f} s = "
          f"{a.drones * a.steps / dt / 1e6:.1f} M env-steps/s (host loop: torch guidance law + one fused step per step)")
    print(f"closest approach to the moving target: median {float(closest.median()):.2f} m, "
          f"within 3 m: {int((closest < 3.0).sum())} of {a.drones}; finite state: {bool(torch.isfinite(drone.position).all())}")


if __name__ == "__main__":
    main()
