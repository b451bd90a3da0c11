"""The float64 CPU restatement (oracle/) against golden vectors captured from the reference itself
(oracle/gen_golden.py: Drone.step /root/reference/src/utils/components.py:220-248 and Racer.step
/root/reference/tests/racer_drone_test.py:95-103).  This pins the oracle; every GPU parity test
then compares the HIP path with the oracle."""
import numpy as np
import pytest

from conftest import load_golden, params_for_golden, racer_params_for_golden
from oracle import oracle

TOL = 1e-12   # float64 restatement vs float64 reference, <= 10 000 steps


def _replay_drone(p, g, dt_fps=None):
    """Step the oracle through the golden's actions, comparing at every snapshot."""
    acts = g["actions"].astype(np.float64)            # [T, n, 4]
    T, n = acts.shape[:2]
    s = oracle.drone_initial_state(n, g["init_position"], g["init_velocity"], g["init_ypr"])
    snaps = [int(x) for x in np.atleast_2d(g["snap_steps"])[0]]   # same for every drone of a file
    done_all = np.zeros((n, T), dtype=np.uint8)
    t0, worst = 0, 0.0
    for k, t1 in enumerate(snaps):
        _, accel, done = oracle.drone_run(p, s, acts[t0:t1], wind=g["wind"])
        done_all[:, t1 - 1] = done          # dense only when the golden snapshots every step (g6)
        t0 = t1
        ref_state, ref_R = g["state"][:, k], g["R"][:, k].reshape(n, 9)
        err = max(np.abs(s[:, 0:6] - ref_state).max(), np.abs(s[:, 6:15] - ref_R).max(),
                  np.abs(s[:, 15:18] - g["prev_rates"][:, k]).max(),
                  np.abs(s[:, 18] - g["prev_thrust"][:, k]).max())
        scale = max(1.0, np.abs(ref_state).max())
        worst = max(worst, err / scale)
        assert err / scale < TOL, f"snapshot {k} (step {t1}): err {err:.3e}"
        assert np.abs(accel - g["accel"][:, k]).max() < 1e-9 * max(1.0, np.abs(g["accel"][:, k]).max())
        assert np.array_equal(done, g["done"][:, t1 - 1])
    return s, done_all, worst


@pytest.mark.parametrize("name", ["g2_sin_4096", "g3_ema_noise", "g4_saturated", "g5_attitude_wind"])
def test_drone_step_matches_reference_1ms(params_1k, name):
    _replay_drone(params_1k, load_golden(name))


def test_config1_zero_sticks_10k(params_1k):
    g = load_golden("g1_zero_10k")
    s, _, _ = _replay_drone(params_1k, g)
    # end state quoted in BASELINE.md / SURVEY.md App. B
    np.testing.assert_allclose(s[0, 0:3], [2.5945630299793589, 0, 206.31842397173011], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(s[0, 3:6], [0.013260768981383429, 0, 20.545675068066696], rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(s[0, 6:15].reshape(3, 3), np.eye(3), atol=1e-15)
    assert abs(s[0, 18] - 31.508531191978875) < 1e-12


def test_default_fps60(params_60):
    _replay_drone(params_60, load_golden("g1b_fps60_sin"))


def test_config0_to_the_letter_default_fps60_zero_sticks_10k(params_60):
    """BASELINE configs[0] as worded: 1 drone, params.yaml defaults (fps = 60, /root/reference/config/params.yaml:2),
    10 000 steps of zero stick input - 166.7 s of flight, the drone climbs to 3.4 km (capture G1 @ fps 60)."""
    g = load_golden("g1_zero_10k_fps60")
    s, done_all, _ = _replay_drone(params_60, g)
    assert abs(float(g["dt"]) - 1 / 60) < 1e-15 and not done_all.any()
    np.testing.assert_allclose(s[0, 0:3], [2.64701726, 0, 3424.66753], rtol=2e-9)
    np.testing.assert_allclose(s[0, 5], 20.5456781, rtol=2e-9)          # terminal climb rate: thrust - weight = drag
    np.testing.assert_allclose(s[0, 6:15].reshape(3, 3), np.eye(3), atol=1e-15)


@pytest.mark.parametrize("k", range(4))
def test_other_drone_types_match_reference(k):
    """Captures G14: four drone types with EVERY parameter Drone.__init__ reads (components.py:86-142) moved away from
    params.yaml - mass, drag coefficients, frame dimensions, max_rates, both transition rates, the motor block of the
    bench report (another thrust cubic), fps 120-2000, gravity 1.62-9.81 - flown with EMA-noise sticks, wind and a
    tilted start.  The host derivation (thrust-curve fit, areas, 5 % / full-throttle forces) must reproduce the
    reference's constants and the oracle the whole trajectory at 1e-12."""
    g = load_golden(f"g14_drone_type_{k}")
    p = params_for_golden(g)
    assert abs(p.dt - float(g["dt"])) < 1e-18 and abs(p.mass - float(g["mass"])) < 1e-15
    np.testing.assert_allclose(p.thrust_poly, g["thrust_poly"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(p.cross_section_areas, g["cross_section_areas"], rtol=1e-15)
    np.testing.assert_allclose([p.min_throttle_in_force, p.max_throttle_in_force],
                               [float(g["min_throttle_in_force"]), float(g["max_throttle_in_force"])], rtol=1e-10)
    s, _, worst = _replay_drone(p, g)
    assert np.isfinite(s).all() and worst < TOL


def test_ground_contact_done_sequence(params_1k):
    g = load_golden("g6_ground")
    _, done_all, _ = _replay_drone(params_1k, g)
    assert np.array_equal(done_all, g["done"])
    assert g["done"][0].any() and not g["done"][0][0]          # falls through the ground
    d2 = g["done"][2]
    assert d2.any() and not d2[-1], "case 2 must recover: done is recomputed, not latched"


def test_return_triple_last_step(params_1k):
    """Drone.step returns (R.T, E(rates used as radians), R_new @ acc) - components.py:247-248."""
    g = load_golden("g3_ema_noise")
    n = g["actions"].shape[1]
    s = oracle.drone_initial_state(n, g["init_position"], g["init_velocity"], g["init_ypr"])
    oracle.drone_run(params_1k, s, g["actions"].astype(np.float64), wind=g["wind"])
    for i in range(n):
        np.testing.assert_allclose(s[i, 6:15].reshape(3, 3).T, g["ret_RT"][i], atol=1e-12)
        r = s[i, 15:18]
        np.testing.assert_allclose(oracle.euler_zyx_matrix(r[0], r[1], r[2]), g["ret_gyro"][i], atol=1e-9)


def test_spot_value_constant_roll(params_1k):
    """SURVEY App. A anchor: roll stick +0.5 for 1 000 steps at 1 ms -> q = [0.17291149, -0.98493737, 0, 0]."""
    s = oracle.drone_initial_state(1, [0, 0, 10.0], [1.0, 0, 0], [0, 0, 0])
    oracle.drone_run(params_1k, s, np.array([[0.5, 0, 0, 0]]), steps=1000)
    np.testing.assert_allclose(s[0, 0:3], [0.9085033271961015, -10.52662258755109, 11.806584453407886], rtol=1e-12)
    q = oracle.matrix_to_quat(s[0, 6:15])[0]
    np.testing.assert_allclose(q, [0.17291149468075978, -0.9849373660325151, 0, 0], atol=1e-12)


@pytest.mark.parametrize("name", ["g7_racer_main", "g8_racer_pid_thrust", "g15_racer_prop7"])
def test_racer_step_matches_reference(params_1k, name):
    """G7: the reference's own scenario; G8: every PID term and a thrust; G15: 7-inch props (another inertia,
    derived by the build's host code), all nine gains non-zero, chirped set-points, a thrust that changes sign."""
    g = load_golden(name)
    p = racer_params_for_golden(g)
    np.testing.assert_allclose(p.racer_inertia, g["inertia"][0], rtol=1e-15)
    if name == "g15_racer_prop7":
        assert abs(p.racer_inertia[0] - 0.5 * (3.5 * 2.54 / 100) ** 2) < 1e-18 and p.racer_inertia[0] > 1.9 * params_1k.racer_inertia[0]
    acts = g["actions"].astype(np.float64)
    s = oracle.racer_initial_state(1)
    t0 = 0
    for k, t1 in enumerate(int(x) for x in g["snap_steps"][0]):
        oracle.racer_run(p, s, acts[t0:t1])
        t0 = t1
        np.testing.assert_allclose(s[0, 10:13], g["omega"][0, k], rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(s[0, 0:3], g["position"][0, k], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(s[0, 3:6], g["velocity"][0, k], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(s[0, 13:16], g["i_error"][0, k], rtol=1e-11, atol=1e-12)
        # quaternion sign is free; compare the rotation it encodes
        qx, qy, qz, qw = s[0, 6:10]
        M = oracle.quat_to_matrix(np.array([qw, qx, qy, qz]))[0]
        np.testing.assert_allclose(M, g["matrix"][0, k], atol=2e-10)


def test_racer_spot_value(params_1k):
    g = load_golden("g7_racer_main")
    np.testing.assert_allclose(g["omega"][0, -1], [-30, -50, 0], atol=1e-9)
    np.testing.assert_allclose(g["inertia"][0], 0.002016125, rtol=1e-12)


def test_ground_plane_contact_matches_reference(params_1k):
    """object_list = [Ground]: per-motor spring inside motor_radius, crash when a motor goes below
    the plane (components.py:198-214, quirk Q5: the crash returns zero collision force)."""
    g = load_golden("g9_ground_contact")
    p = params_1k.replace(ground=True)
    _, done_all, _ = _replay_drone(p, g)
    assert np.array_equal(done_all, g["done"])
    assert g["done"][2].any() and not g["done"][0].any()
    zmin = g["state"][:, :, 2].min(axis=1)
    assert 0 < zmin[0] < 0.1 and 0 < zmin[1] < 0.1, "cases 0/1 must enter the spring zone without crashing"
    # without the flag the same inputs fall through the plane: the contact force is what differs
    s = oracle.drone_initial_state(4, g["init_position"], g["init_velocity"], g["init_ypr"])
    oracle.drone_run(params_1k, s, g["actions"].astype(np.float64))
    assert s[0, 2] < 0 < g["state"][0, -1, 2]


def _g10_objects(g, t):
    objs = [tuple(o) for o in g["objects"]]
    tp = g["target_positions"][t]
    objs[0] = (2, tp[0], tp[1], tp[2], float(g["target_radius"]), 0.0)    # the Target moves before each step
    return tuple(objs)


def test_object_list_collisions_match_reference(params_1k):
    """object_list = [moving Target, Cylinder, Cylinder, Ground] (simulator.py:85-87 order):
    cylinder side / top springs, sphere contact, ground + cylinder together, and crashes that keep
    the forces of earlier objects (components.py:198-214)."""
    g = load_golden("g10_objects")
    acts = g["actions"].astype(np.float64)
    T, n = acts.shape[:2]
    s = oracle.drone_initial_state(n, g["init_position"], g["init_velocity"], g["init_ypr"])
    for t in range(T):
        p = params_1k.replace(objects=_g10_objects(g, t))
        _, accel, done = oracle.drone_run(p, s, acts[t:t + 1])
        assert np.array_equal(done, g["done"][:, t]), t
        assert np.abs(s[:, 0:6] - g["state"][:, t]).max() < 1e-11, t
        assert np.abs(s[:, 6:15] - g["R"][:, t].reshape(n, 9)).max() < 1e-12
    assert g["done"][1].any() and g["done"][5].any() and not g["done"][0].any()


def test_raised_objects_match_reference(params_1k):
    """Capture G16: object_list = [Ground, Cylinder spanning z = 1.2 .. 3.2, standing Target, small raised Cylinder].
    Pins what G10's cylinders (standing on z = 0) cannot show - Cylinder.calculate_normal tests the height RELATIVE to
    the base against the ABSOLUTE band (components.py:718-720), so a side contact at z = 1.8 is pushed DOWN while one
    at z = 2.8 is pushed out - plus the rim from below, the top (no spring: crash on entering the band), a sphere that
    does not move, and the list order with Ground first."""
    g = load_golden("g16_objects_raised")
    acts = g["actions"].astype(np.float64)
    T, n = acts.shape[:2]
    p = params_1k.replace(objects=tuple(tuple(o) for o in g["objects"]))
    s = oracle.drone_initial_state(n, g["init_position"], g["init_velocity"], g["init_ypr"])
    for t in range(T):
        _, accel, done = oracle.drone_run(p, s, acts[t:t + 1])
        assert np.array_equal(done, g["done"][:, t]), t
        assert np.abs(s[:, 0:6] - g["state"][:, t]).max() < 1e-11, t
        assert np.abs(s[:, 6:15] - g["R"][:, t].reshape(n, 9)).max() < 1e-12
    assert ((g["deviation_from_free_flight"] > 0.5) | g["done"].any(axis=1)).all(), "every drone of the capture must meet an object"
    # the quirk itself: drone 0 (side contact at relative height 0.6) is driven DOWN
    # (vz -3 m/s, forward speed untouched, until a motor enters the cylinder); drone 1 (relative 1.6) bounces back at its height
    assert g["state"][0, 700, 5] < -2.5 and g["state"][0, 700, 3] > 0.44 and g["done"][0, 701]
    assert abs(g["state"][1, -1, 2] - 2.8) < 0.02 and g["state"][1, -1, 3] < -0.4
    assert g["done"][3].any() and not g["done"][1].any() and not g["done"][2].any() and not g["done"][4].any()


def _cpid_params(p, g):
    return p.replace(mode=1, racer_pid=g["pid"], racer_pid_variant=1, pid_integral_clip=float(g["clips"][0]),
                     pid_min_output=float(g["clips"][1]), pid_max_output=float(g["clips"][2]),
                     pid_derivative_transition_rate=float(g["clips"][3]))


def test_components_pid_matches_reference_class():
    """a16: PID.__call__ (/root/reference/src/utils/components.py:43-54) on the seeded sequences of G11,
    which reach the integral clip, the 0.99 leak, the +-1 derivative clip, the derivative low-pass
    and both output clips (asserted here so the fixture cannot silently lose its coverage)."""
    g = load_golden("g11_components_pid")
    for c in range(g["gains"].shape[0]):
        out, integ, der, err = oracle.pid_run(g["gains"][c], g["current"][c], g["target"][c])
        for got, key in ((out, "out"), (integ, "integral"), (der, "derivative"), (err, "error")):
            assert np.abs(got - g[key][c]).max() <= TOL * max(1.0, np.abs(g[key][c]).max()), (c, key)
    gains = g["gains"]
    assert (np.abs(g["integral"][1]) >= gains[1, 4] - 1e-15).any(), "integral clip not reached"
    assert (g["out"][1] <= gains[1, 5]).any() and (g["out"][1] >= gains[1, 6]).any(), "output clips not reached"
    raw_d = np.abs(np.diff(g["error"][1])).max() / gains[1, 3]
    assert raw_d > 1.0, "derivative clip not exercised"
    assert ((g["out"][2] > gains[2, 5]) & (g["out"][2] < gains[2, 6])).all(), "case 2 must stay unclipped"
    # leak: with a constant error e the integral settles at e*dt/(1-0.99) unless clipped first
    assert abs(g["integral"][3][-1] - min(1.0 * gains[3, 3] / 0.01, gains[3, 4])) < 1e-6


def test_racer_with_components_pid_matches_reference():
    """Racer.step with components.PID objects in its pid dict (both classes are the reference's own code,
    the harness adapts the call shape): the oracle's racer_pid_variant = 1."""
    g = load_golden("g12_racer_components_pid")
    p = _cpid_params(load_params_1k(), g)
    ref = oracle.racer_initial_state(1)
    prev = 0
    for k, t in enumerate(np.asarray(g["snap_steps"]).reshape(-1)):
        oracle.racer_run(p, ref, g["actions"][prev:int(t)].astype(np.float64))
        prev = int(t)
        for sl, key in ((slice(0, 3), "position"), (slice(3, 6), "velocity"), (slice(6, 10), "quat_xyzw"),
                        (slice(10, 13), "omega"), (slice(13, 16), "i_error"), (slice(20, 23), "prev_derivative")):
            assert np.abs(ref[0, sl] - g[key][0, k]).max() < TOL, (k, key)


def load_params_1k():
    from fpyv_amd import load_params
    return load_params(fps=1000)


def test_guidance_override_matches_reference(params_1k):
    """Drone.step(..., rotation_matrix=R, thrust_force=f) (components.py:230-232) on G13: override switched on
    and off mid-flight (case 0), on every step (case 1), on every step above a Ground object (case 2)."""
    g = load_golden("g13_guidance_override")
    acts = g["actions"].astype(np.float64)
    T, n = acts.shape[:2]
    assert np.isnan(g["thrust_force"][:, 0]).sum() == 300 and not np.isnan(g["thrust_force"][:, 1:]).any()
    for i in range(n):
        p = params_1k.replace(objects=[(0, 0, 0, 0, 0, 0)]) if g["ground_case"][i] else params_1k
        s = oracle.drone_initial_state(1, g["init_position"][i], g["init_velocity"][i], g["init_ypr"][i])[0]
        states, accel, done = oracle.drone_run_guided(p, s, acts[:, i], g["rotation_override"][:, i], g["thrust_force"][:, i])
        assert np.abs(states[:, 0:6] - g["state"][i]).max() < TOL * max(1.0, np.abs(g["state"][i]).max())
        assert np.abs(states[:, 6:15] - g["R"][i].reshape(T, 9)).max() < TOL
        assert np.abs(states[:, 15:18] - g["prev_rates"][i]).max() < TOL * 200
        assert np.abs(states[:, 18] - g["prev_thrust"][i]).max() < TOL * 100
        assert np.abs(accel - g["accel"][i]).max() < 1e-9 * max(1.0, np.abs(g["accel"][i]).max())
        assert np.array_equal(done, g["done"][i])
    # the override really bites: on an overridden step the attitude after the step is R_override turned by the
    # step's double increment, i.e. within the increment's size of the commanded matrix and far from the free flight
    i, t = 1, 300
    assert np.abs(g["R"][i, t] - g["rotation_override"][t, i]).max() < 0.02
