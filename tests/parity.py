"""Shared comparison helpers: fp32 SoA state (HIP kernel / lane model) vs the float64 oracle state."""
import numpy as np

from oracle import oracle

# BASELINE.json north_star: "within 1e-5 relative on position/quaternion after 1000 steps"
REL_TOL = 1e-5


def soa_vs_oracle(soa: np.ndarray, ref: np.ndarray, n: int):
    """soa: [14, ld] fp32 (p, v, q wxyz, rates, thrust); ref: [n, 19] float64 oracle state.
    Returns dict of error measures:
      pos_rel   max_i |dp_i| / |p_i|          (vector relative error per drone)
      pos_comp  max_i,c |dp_ic| / max(|p_ic|, 1)   (per component with a 1 m floor)
      vel_rel   max_i |dv_i| / max(|v_i|, 1)
      quat_abs  max_i,c |dq| with the sign of q aligned (|q| = 1, so absolute = relative)
      rot_frob  max_i ||R(q) - R_ref||_F
    """
    p = soa[0:3, :n].T.astype(np.float64)
    v = soa[3:6, :n].T.astype(np.float64)
    q = soa[6:10, :n].T.astype(np.float64)
    pr, vr, Rr = ref[:, 0:3], ref[:, 3:6], ref[:, 6:15].reshape(n, 3, 3)
    qr = oracle.matrix_to_quat(Rr)
    sign = np.sign(np.sum(q * qr, axis=1, keepdims=True))
    sign[sign == 0] = 1
    dq = np.abs(q * sign - qr).max()
    Rq = oracle.quat_to_matrix(q)
    out = dict(
        pos_rel=(np.linalg.norm(p - pr, axis=1) / np.linalg.norm(pr, axis=1)).max(),
        pos_comp=(np.abs(p - pr) / np.maximum(np.abs(pr), 1.0)).max(),
        vel_rel=(np.linalg.norm(v - vr, axis=1) / np.maximum(np.linalg.norm(vr, axis=1), 1.0)).max(),
        quat_abs=dq,
        rot_frob=np.linalg.norm((Rq - Rr).reshape(n, 9), axis=1).max(),
        rates_abs=np.abs(soa[10:13, :n].T - ref[:, 15:18]).max(),
        thrust_rel=(np.abs(soa[13, :n] - ref[:, 18]) / np.maximum(np.abs(ref[:, 18]), 1.0)).max(),
        qnorm=np.abs(np.linalg.norm(q, axis=1) - 1).max(),
    )
    return out


def assert_parity(err, tol=REL_TOL, what=""):
    assert err["pos_rel"] <= tol, f"{what} position rel err {err['pos_rel']:.3e} > {tol}"
    assert err["pos_comp"] <= tol, f"{what} position per-component err {err['pos_comp']:.3e} > {tol}"
    assert err["quat_abs"] <= tol, f"{what} quaternion err {err['quat_abs']:.3e} > {tol}"
    assert err["qnorm"] <= 5e-7, f"{what} |q|-1 = {err['qnorm']:.3e}"


def assert_parity_random_type(soa, ref, n, p, tol=REL_TOL, what=""):
    """The 1e-5 bar for drone types and initial poses drawn at random: the position error is taken relative to
    max(|p|, distance travelled from the initial position) - a flight that happens to pass the origin must not turn
    1.5e-5 m of error on 20 m of travel into a "relative" error above the bar - the quaternion error as is."""
    import numpy as np
    err = soa_vs_oracle(soa, ref, n)
    pos = soa[0:3, :n].T.astype(np.float64)
    scale = np.maximum(np.linalg.norm(ref[:, 0:3], axis=1), np.linalg.norm(ref[:, 0:3] - np.asarray(p.init_position), axis=1))
    rel = (np.linalg.norm(pos - ref[:, 0:3], axis=1) / np.maximum(scale, 1.0)).max()
    assert rel <= tol, f"{what} position err / max(|p|, path) = {rel:.3e} > {tol}"
    assert err["quat_abs"] <= tol, f"{what} quaternion err {err['quat_abs']:.3e} > {tol}"
    assert err["qnorm"] <= 5e-7 and err["vel_rel"] <= tol, f"{what} {err}"
    return rel


def random_drone_params(base, rng):
    """A physically plausible drone type far from the packaged defaults: the parity claims must not depend on
    params.yaml's numbers.  Motors stay an X frame or become a rectangular one (the two-height ground flag of
    the k-step kernels must switch itself off for the latter)."""
    import numpy as np
    mass = rng.uniform(0.25, 2.5)
    arm = rng.uniform(0.06, 0.3)
    if rng.random() < 0.5:
        ang = np.pi / 4 + np.arange(4) * np.pi / 2
        motors = np.stack([arm * np.cos(ang), arm * np.sin(ang)], axis=1)
    else:
        lx, ly = arm * rng.uniform(0.6, 1.0), arm * rng.uniform(0.6, 1.0)
        motors = np.array([[lx, ly], [-lx, ly], [-lx, -ly], [lx, -ly]])
    hover = mass * 9.81
    c1 = hover * rng.uniform(0.025, 0.05)                # thrust over throttle percent: hover at 20-40 %
    poly = np.array([-c1 * rng.uniform(1e-5, 8e-5), c1 * rng.uniform(0.0, 0.02), c1, -rng.uniform(0.0, 0.1)])
    return base.replace(
        dt=1.0 / rng.choice([250.0, 500.0, 1000.0, 2000.0]), mass=mass, gravity=rng.uniform(3.7, 11.0),
        max_rates=rng.uniform(90.0, 900.0), rates_transition_rate=rng.uniform(0.2, 1.0),
        thrust_transition_rate=rng.uniform(0.2, 1.0), thrust_poly=poly,
        drag_coefficients=rng.uniform(0.4, 2.5, 3), cross_section_areas=rng.uniform(0.004, 0.1, 3),
        air_density=rng.uniform(0.9, 1.3), motor_xy=motors,
        init_position=np.array([rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(5, 40)]),
        init_velocity=rng.uniform(-3, 3, 3), init_orientation_deg=rng.uniform([-180, -60, -180], [180, 60, 180]))
