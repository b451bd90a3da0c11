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
