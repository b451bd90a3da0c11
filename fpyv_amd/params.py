"""Parameter loading for the batched FPV stepper.

Host-side, init-time only.  Restates (does not import) what the reference does when a `Drone` is
constructed:

* YAML schema: /root/reference/config/params.yaml:1-3 (simulator.fps, simulator.gravity) and
  :38-51 (drone.*).  Derived constants follow /root/reference/src/utils/components.py:92-100
  (dt = 1/fps, mass g->kg, dimensions cm->m, cross-section areas) and :120-125 (motor positions).
* Motor bench report -> per-block tables: /root/reference/src/utils/flight_time_calculator.py:16-40.
* Thrust curve: cubic least-squares fit through (0,0) + the block's points,
  /root/reference/src/utils/flight_time_calculator.py:43-52, thrust_N = n_motors*g/1000*gravity
  (/root/reference/src/utils/components.py:134).  The reference re-fits on every step; the fit is
  a pure function of the table, so it is done once here.

All values are float64; the C ABI receives them as doubles and the HIP side narrows to fp32.
"""
from __future__ import annotations

import copy
import csv
import dataclasses
import math
import os
from typing import Any, Dict, List, Optional, Sequence

import numpy as np
import yaml

_DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
DEFAULT_PARAMS_PATH = os.path.join(_DATA_DIR, "stepper_defaults.yaml")

MODE_DRONE = 0   # Drone.step arithmetic (components.py:220-248)
MODE_RACER = 1   # Racer.step arithmetic (tests/racer_drone_test.py:95-103)
_MODE_NAMES = {"drone": MODE_DRONE, "racer": MODE_RACER}


def read_motor_test_report(path: str) -> List[Dict[str, np.ndarray]]:
    """Parse a motor bench report into blocks of {'throttle': %, 'thrust': grams}.

    Accepts the build's normalised 3-column table (`block,throttle_pct,thrust_g`, '#' comments) and
    the raw T-Motor export the reference reads (10 columns, '50%' throttle cells, decimal commas in
    thrust cells; optional header row starting with 'Type').  Raw files are split into blocks that
    end at each 100 % row, like flight_time_calculator.py:34-39.
    """
    with open(path, encoding="utf-8") as f:
        lines = [ln for ln in f if ln.strip() and not ln.lstrip().startswith("#")]
    rows = list(csv.reader(lines))
    if not rows:
        raise ValueError(f"motor test report {path!r} is empty")
    blocks: List[Dict[str, List[float]]] = []
    if rows[0][0].strip() == "block":          # normalised schema
        for r in rows[1:]:
            b = int(r[0])
            while len(blocks) <= b:
                blocks.append({"throttle": [], "thrust": []})
            blocks[b]["throttle"].append(float(r[1]))
            blocks[b]["thrust"].append(float(r[2]))
    else:                                       # raw vendor export
        if rows[0][0].strip() == "Type":
            rows = rows[1:]
        cur: Dict[str, List[float]] = {"throttle": [], "thrust": []}
        for r in rows:
            thr = float(r[2].replace("%", ""))
            cur["throttle"].append(thr)
            cur["thrust"].append(float(r[3].replace(",", ".")))
            if thr == 100.0:
                blocks.append(cur)
                cur = {"throttle": [], "thrust": []}
        if cur["throttle"]:
            blocks.append(cur)
    return [{k: np.asarray(v, dtype=np.float64) for k, v in b.items()} for b in blocks]


def fit_through_origin(x: Sequence[float], y: Sequence[float], degree: int = 3) -> np.ndarray:
    """Least-squares polynomial through the points with (0, 0) prepended; highest power first.

    Same estimator as flight_time_calculator.py:43-52 (`np.polyfit` on the augmented points).
    """
    xa = np.append(0.0, np.asarray(x, dtype=np.float64))
    ya = np.append(0.0, np.asarray(y, dtype=np.float64))
    return np.polyfit(xa, ya, degree)


def ypr_to_quat(roll_deg: float, pitch_deg: float, yaw_deg: float) -> np.ndarray:
    """Quaternion (w,x,y,z) of Rz(yaw)·Ry(pitch)·Rx(roll), angles in degrees.

    Drone.reset builds R = euler_angles_to_rotation_matrix(*deg2rad(ypr)) and, despite the
    parameter name, consumes the triple as (roll, pitch, yaw) (components.py:150-154,
    helper_functions.py:39-43).
    """
    r, p, y = (math.radians(a) * 0.5 for a in (roll_deg, pitch_deg, yaw_deg))
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    return np.array([cy * cp * cr + sy * sp * sr,
                     cy * cp * sr - sy * sp * cr,
                     cy * sp * cr + sy * cp * sr,
                     sy * cp * cr - cy * sp * sr], dtype=np.float64)


@dataclasses.dataclass
class DroneParams:
    """Everything the per-drone step needs, as float64 scalars / small arrays."""
    mode: int = MODE_DRONE
    dt: float = 1.0 / 60.0
    gravity: float = 9.81
    mass: float = 0.75
    max_rates: float = 200.0
    rates_transition_rate: float = 0.7
    thrust_transition_rate: float = 0.5
    thrust_poly: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros(4))   # c3..c0, x = throttle %
    inverse_thrust_poly: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros(4))
    drag_coefficients: np.ndarray = dataclasses.field(default_factory=lambda: np.array([1.8, 1.8, 1.2]))
    cross_section_areas: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros(3))
    air_density: float = 1.2225
    motor_xy: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros((4, 2)))
    init_position: np.ndarray = dataclasses.field(default_factory=lambda: np.array([0.0, 0.0, 10.0]))
    init_velocity: np.ndarray = dataclasses.field(default_factory=lambda: np.array([1.0, 0.0, 0.0]))
    init_orientation_deg: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros(3))
    ceiling: float = math.inf
    goal: np.ndarray = dataclasses.field(default_factory=lambda: np.array([0.0, 0.0, 10.0]))
    min_throttle_in_force: float = 0.0
    max_throttle_in_force: float = 0.0
    # rate-PID / torque loop (tests/racer_drone_test.py:68-83)
    racer_mass: float = 0.5
    racer_inertia: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros(3))
    racer_pid: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros((3, 3)))
    racer_velocity_damping: float = 0.9
    racer_omega_dt: bool = False    # False = rotate by omega per step as the reference writes it
    # rate-loop semantics: 0 = racer_drone_test.PID.step (:22-32), 1 = components.PID.__call__ (components.py:43-54)
    racer_pid_variant: int = 0
    pid_integral_clip: float = 1.0                 # components.py:16 defaults
    pid_min_output: float = 0.3
    pid_max_output: float = 1.0
    pid_derivative_transition_rate: float = 0.5
    # ground plane in object_list (components.py:198-214, :121; Ground.calculate_distance = z, :674-677)
    ground: bool = False
    motor_radius: float = 0.1
    ground_spring: float = 100.0
    ground_damping: float = 0.0
    # in-kernel stick noise (tests/noise_smooth_test.py:5-11): x_s <- (1 - tau) x_s + tau N(0,1)
    noise_transition: float = 0.1
    noise_gain: float = 1.0
    # drone.force_multiplier_pid of params.yaml (:55-62): kwargs of the guidance PID (components.py:143-145); its
    # min_output / max_output are replaced by the 5 % / full throttle forces when the drone builds the controller
    force_multiplier_pid: Dict[str, float] = dataclasses.field(default_factory=lambda: dict(
        kP=0.1, kI=2.0, kD=0.05, integral_clip=100.0, min_output=0.05, max_output=40.0, derivative_transition_rate=0.2))
    # ordered object_list for the collision pass (components.py:198-214): tuples
    # (type, x, y, z, radius, height) with type 0 = Ground, 1 = Cylinder, 2 = Target sphere; max 8
    objects: tuple = ()

    @property
    def init_quat(self) -> np.ndarray:
        return ypr_to_quat(*self.init_orientation_deg)

    def thrust_from_stick(self, throttle: Any) -> Any:
        """throttle stick in [-1, 1] -> total thrust [N] (components.py:136)."""
        return np.polyval(self.thrust_poly, 100.0 * (np.asarray(throttle, dtype=np.float64) + 1.0) / 2.0)

    def stick_from_thrust(self, thrust: Any) -> Any:
        """total thrust [N] -> throttle stick, clipped to [-1, 1] (components.py:137)."""
        return np.clip(np.polyval(self.inverse_thrust_poly, thrust) / 100.0 * 2.0 - 1.0, -1.0, 1.0)

    def replace(self, **kw) -> "DroneParams":
        return dataclasses.replace(copy.deepcopy(self), **kw)


def _resolve_report_path(raw: str, yaml_dir: str) -> str:
    """The reference YAML carries an absolute Windows path (params.yaml:39-40); fall back to the
    file name next to the YAML, then to the packaged table."""
    cands = [raw, os.path.join(yaml_dir, raw)]
    base = raw.replace("\\", "/").split("/")[-1]
    cands += [os.path.join(yaml_dir, base), os.path.join(_DATA_DIR, base),
              os.path.join(_DATA_DIR, "f80_thrust_table.csv")]
    for c in cands:
        if os.path.isfile(c):
            return c
    raise FileNotFoundError(f"motor test report not found (tried {cands})")


def params_from_dict(cfg: Dict[str, Any], yaml_dir: str = _DATA_DIR, mode: Any = "drone",
                     fps: Optional[float] = None) -> DroneParams:
    """Build DroneParams from a params.yaml-shaped dict.  The dict is not modified
    (the reference constructor mutates its argument, components.py:143-144; this one does not)."""
    sim, drone = cfg["simulator"], cfg["drone"]
    st = cfg.get("stepper", {}) or {}
    racer = st.get("racer", {}) or {}
    gravity = float(sim["gravity"])
    fps_v = float(fps if fps is not None else sim["fps"])
    n_motors = int(st.get("n_motors", 4))
    if n_motors != 4:
        raise ValueError("the stepper is built for 4 motors (components.py:120)")

    dims = np.asarray(drone["dimensions"], dtype=np.float64) / 100.0          # components.py:99
    areas = np.array([dims[1] * dims[2], dims[0] * dims[2], dims[0] * dims[1]])  # components.py:100

    radius = float(st.get("arm_radius_inch", 5)) * 2.54 / 100                   # components.py:122
    t = np.linspace(0, 2 * np.pi, n_motors + 1)[:-1]                            # components.py:123
    t = t + (t[1] - t[0]) / 2                                                   # components.py:124
    motor_xy = radius * np.stack([np.cos(t), np.sin(t)], axis=1)                # components.py:125

    report = read_motor_test_report(_resolve_report_path(str(drone["motor_test_report_path"]), yaml_dir))
    block = report[int(drone["motor_test_report_idx"])]
    thrust_n = n_motors * block["thrust"] / 1000 * gravity                      # components.py:134
    poly = fit_through_origin(block["throttle"], thrust_n)
    inv_poly = fit_through_origin(thrust_n, block["throttle"])                  # components.py:137

    prop_r = (float(racer.get("prop_size_inch", 5)) / 2) * 2.54 / 100           # racer_drone_test.py:70
    racer_mass = float(racer.get("mass", 0.5))
    pid = racer.get("pid", {"roll": [2, 0, 0], "pitch": [2, 0, 0], "yaw": [0.1, 0, 0]})

    p = DroneParams(
        mode=_MODE_NAMES[mode] if isinstance(mode, str) else int(mode),
        dt=1 / fps_v,                                                           # components.py:96
        gravity=gravity,
        mass=float(drone["mass"]) / 1000,                                       # components.py:97
        max_rates=float(drone["max_rates"]),
        rates_transition_rate=float(drone["rates_transition_rate"]),
        thrust_transition_rate=float(drone["thrust_transition_rate"]),
        thrust_poly=poly,
        inverse_thrust_poly=inv_poly,
        drag_coefficients=np.asarray(drone["drag_coefficients"], dtype=np.float64),
        cross_section_areas=areas,
        air_density=float(st.get("air_density", 1.2225)),                       # kinematics.py:33
        motor_xy=motor_xy,
        init_position=np.asarray(drone["initial_position"], dtype=np.float64),
        init_velocity=np.asarray(drone["initial_velocity"], dtype=np.float64),
        init_orientation_deg=np.asarray(drone["initial_orientation"], dtype=np.float64),
        ceiling=float(st.get("ceiling", math.inf)),
        goal=np.asarray(st.get("goal", drone["initial_position"]), dtype=np.float64),
        racer_mass=racer_mass,
        racer_inertia=racer_mass * prop_r ** 2 * np.ones(3),                    # racer_drone_test.py:83
        racer_pid=np.asarray([pid["roll"], pid["pitch"], pid["yaw"]], dtype=np.float64),
        racer_velocity_damping=float(racer.get("velocity_damping", 0.9)),
        racer_omega_dt=bool(racer.get("omega_dt", False)),
        racer_pid_variant=int(racer.get("pid_variant", 0)),
        pid_integral_clip=float(racer.get("integral_clip", 1.0)),
        pid_min_output=float(racer.get("min_output", 0.3)),
        pid_max_output=float(racer.get("max_output", 1.0)),
        pid_derivative_transition_rate=float(racer.get("derivative_transition_rate", 0.5)),
        ground=bool(st.get("ground", False)),
        motor_radius=float(st.get("motor_radius", 0.1)),
        ground_spring=float(st.get("ground_spring", 100.0)),
        ground_damping=float(st.get("ground_damping", 0.0)),
    )
    if "force_multiplier_pid" in drone:                                          # params.yaml:55-62
        p.force_multiplier_pid = {k: float(v) for k, v in drone["force_multiplier_pid"].items()}
    # 5 % throttle floor / full throttle, components.py:139-142
    p.min_throttle_in_force = float(p.thrust_from_stick(-1 + 5 / 100 * 2))
    p.max_throttle_in_force = float(p.thrust_from_stick(1.0))
    if not p.min_throttle_in_force > 0:
        raise ValueError("The minimum throttle is below zero. This is not possible.")  # components.py:141
    return p


def load_params(path: Optional[str] = None, mode: Any = "drone", fps: Optional[float] = None,
                **overrides: Any) -> DroneParams:
    """Read a params.yaml (reference schema) and derive the step constants.

    `fps` overrides simulator.fps (the 1 ms configuration is fps=1000).  Extra keyword arguments
    replace DroneParams fields after derivation.
    """
    path = path or DEFAULT_PARAMS_PATH
    with open(path, encoding="utf-8") as f:
        cfg = yaml.safe_load(f)
    p = params_from_dict(cfg, yaml_dir=os.path.dirname(os.path.abspath(path)), mode=mode, fps=fps)
    return p.replace(**overrides) if overrides else p
