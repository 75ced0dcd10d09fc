"""The plant (rows a26 / a27 of SURVEY.md section 8) against CLOSED FORMS — not against its twin.

Gazebo 11 / ODE and the RotorS motor-model plugin cannot run here, so no fixture can pin `plant_step` / `rotor_filter` /
`platform_contact` (csrc/dql_device.hpp) or their oracle twin (oracle/dql_oracle.c: motor_and_body).  What can be checked
independently of both: the formulas the reference's sources state, evaluated in float64 numpy from the constants those sources hold.
Every test below runs the same assertions on the CPU oracle (float64 and float32) and — under `-m gpu` — on the HIP operator
`dql_plant_run` through the C ABI.  References (paths under /root/reference/src):
  rotors_simulator/rotors_gazebo_plugins/src/gazebo_motor_model.cpp:358-364   commanded speed limited to max_rot_velocity
  .../gazebo_motor_model.cpp:434-452   thrust = k_f omega^2 along the rotor axis
  .../gazebo_motor_model.cpp:453-469   rotor drag  -|omega| c_d v_perp  (world frame, v_perp = v - (v . n) n)
  .../gazebo_motor_model.cpp:470-482   drag torque -turning_direction * thrust * k_m about the rotor axis
  .../gazebo_motor_model.cpp:484-489   rolling moment -|omega| c_r v_perp
  .../include/rotors_gazebo_plugins/common.h:147-183   first-order filter: x+ = a x + (1 - a) u, a = exp(-dt / tau_up|down)
  rotors_simulator/rotors_description/urdf/hummingbird.xacro:29-42,69-155   vehicle constants, rotor positions and directions
  dql_multirotor_landing/worlds/basic.world:36,64-70   g = 9.8, dt = 0.002
  dql_multirotor_landing/urdf/moving_platform.urdf   platform top 0.455 m, 1 x 1 m
Parity vs Gazebo itself stays unpinned; these pin the restatement vs the cited formulas.
"""
import numpy as np
import pytest

from dql_multirotor_landing_amd.config import DqlConfig, F32, F64

# ---- constants as the reference's files hold them (NOT read from config.py: the config is part of what is under test) ----
M = 0.68 + 4 * 0.009 + 1e-5      # hummingbird.xacro:29,32 + the odometry sensor's 1e-5 kg link
KF, KM = 8.54858e-06, 0.016      # :36-37
ARM, ROTOR_Z = 0.17, 0.01        # :33-34
TAU_UP, TAU_DOWN = 0.0125, 0.025 # :38-39
OMEGA_MAX = 838.0                # :40
C_DRAG, C_ROLL = 8.06428e-05, 1e-06  # :41-42
G, DT = 9.8, 0.002               # basic.world:36, 64-70
# rotor i: position in the body frame and turning direction (+1 ccw, -1 cw): hummingbird.xacro:69-155
ROTOR_POS = np.array([[ARM, 0, ROTOR_Z], [0, ARM, ROTOR_Z], [-ARM, 0, ROTOR_Z], [0, -ARM, ROTOR_Z]])
ROTOR_DIR = np.array([-1, +1, -1, +1])
HOVER = np.sqrt(M * G / (4 * KF))
PLATFORM_TOP, HALF, BOTTOM = 0.455, 0.5 + 0.05, 0.06


def _backends():
    out = [pytest.param(("oracle", F64), id="oracle-f64"), pytest.param(("oracle", F32), id="oracle-f32"),
           pytest.param(("hip", F32), id="hip-f32", marks=pytest.mark.gpu), pytest.param(("hip", F64), id="hip-f64", marks=pytest.mark.gpu)]
    return out


@pytest.fixture(params=_backends())
def plant(request):
    kind, dtype = request.param
    cfg = DqlConfig(dtype=dtype)
    if kind == "oracle":
        from oracle import oracle as orc
        orc.build()
        run = lambda init, cmd: orc.plant_run(cfg, init, cmd)
    else:
        import __graft_entry__ as g
        g.build_hip()
        from dql_multirotor_landing_amd import ops
        run = lambda init, cmd: ops.plant_run(cfg, init, cmd)
    run.cfg = cfg
    run.tol = 1e-10 if dtype == F64 else 2e-6   # relative, per tick
    run.f64 = dtype == F64
    return run


def init_state(p=(0, 0, 4.0), v=(0, 0, 0), q=(1, 0, 0, 0), w=(0, 0, 0), om=(HOVER,) * 4, mp=(50.0, 50.0, 0.0, 0.0)):
    return np.array([[*p, *v, *q, *w, *om, *mp]], dtype=np.float64)


def const_cmd(om, n):
    return np.tile(np.asarray(om, dtype=np.float64), (1, n, 1))


P, V, Q, W, OM, MPX, MPY, CONTACT = slice(0, 3), slice(3, 6), slice(6, 10), slice(10, 13), slice(13, 17), 17, 18, 19


def test_config_holds_the_reference_constants():
    c = DqlConfig()
    assert (c.mass, c.k_f, c.k_m, c.arm_length, c.rotor_z) == (M, KF, KM, ARM, ROTOR_Z)
    assert (c.tau_up, c.tau_down, c.rotor_max, c.c_drag, c.c_roll, c.gravity, c.dt) == (TAU_UP, TAU_DOWN, OMEGA_MAX, C_DRAG, C_ROLL, G, DT)
    assert (c.mp_top_z, c.mp_half_x, c.mp_half_y, c.drone_bottom) == (PLATFORM_TOP, HALF, HALF, BOTTOM)


def test_rotor_step_response_up_down_and_speed_limit(plant):
    """common.h:147-183: omega_k = ref + (omega_0 - ref) a^k, a = exp(-dt / tau_up) while accelerating, exp(-dt / tau_down) while
    decelerating; gazebo_motor_model.cpp:358-364: ref = min(cmd, max_rot_velocity)."""
    n = 60
    cmd = np.concatenate([const_cmd([600, 700, 2000, 300], n), const_cmd([200, 100, 0, 300], n)], axis=1)
    out = plant(init_state(om=(300, 300, 300, 300)), cmd)[0]
    a_up, a_dn = np.exp(-DT / TAU_UP), np.exp(-DT / TAU_DOWN)
    k = np.arange(1, n + 1)
    for i, (ref_up, ref_dn) in enumerate([(600, 200), (700, 100), (OMEGA_MAX, 0), (300, 300)]):
        up = ref_up + (300 - ref_up) * a_up ** k
        top = up[-1]
        dn = ref_dn + (top - ref_dn) * a_dn ** k
        np.testing.assert_allclose(out[:n, 13 + i], up, rtol=plant.tol * 40)
        np.testing.assert_allclose(out[n:, 13 + i], dn, rtol=plant.tol * 40, atol=1e-4 if not plant.f64 else 1e-9)
    # the time constant read back from the response: after 25 ticks = 4 tau_up the step is 1 - e^-4 complete
    assert abs((out[24, 13] - 300) / 300 - (1 - np.exp(-4.0))) < 1e-5


def test_hover_equilibrium_holds_vertical_speed(plant):
    """four rotors at sqrt(m g / 4 k_f) (gazebo_motor_model.cpp:441-452 thrust, basic.world:36 gravity): 5 s without drift"""
    n = 2500
    out = plant(init_state(), const_cmd([HOVER] * 4, n))[0]
    lim = 1e-6 if plant.f64 else 2e-3  # float32: thrust and weight differ by rounding (1e-7 g), integrated over 5 s
    assert np.abs(out[:, 5]).max() <= lim
    assert np.abs(out[:, 2] - 4.0).max() <= (1e-5 if plant.f64 else 1e-2)
    assert np.abs(out[:, W]).max() == 0.0 and np.abs(out[:, 3:5]).max() == 0.0
    np.testing.assert_allclose(out[:, Q], np.tile([1.0, 0, 0, 0], (n, 1)), atol=0)


def test_thrust_gravity_and_free_fall(plant):
    """a_z = k_f sum(omega^2) / m - g; semi-implicit Euler: v += dt a, p += dt v(new).  Rotors off: v_k = -g dt k,
    p_k = p_0 - g dt^2 k (k + 1) / 2 exactly."""
    om = np.array([400.0, 420.0, 380.0, 410.0])
    out = plant(init_state(om=om), const_cmd(om, 1))[0]
    az = KF * np.sum(om ** 2) / M - G
    np.testing.assert_allclose(out[0, 5], DT * az, rtol=plant.tol * 10)
    # float32 holds 4.0 - 8e-6 to half an ulp of 4.0 (2.4e-7)
    np.testing.assert_allclose(out[0, 2] - 4.0, DT * DT * az, rtol=1e-9, atol=0 if plant.f64 else 2.4e-7)
    n = 200
    out = plant(init_state(om=(0, 0, 0, 0)), const_cmd([0] * 4, n))[0]
    k = np.arange(1, n + 1)
    np.testing.assert_allclose(out[:, 5], -G * DT * k, rtol=plant.tol * 300)
    np.testing.assert_allclose(out[:, 2], 4.0 - G * DT * DT * k * (k + 1) / 2, rtol=plant.tol * 300)


def test_roll_pitch_torque_sign_and_magnitude(plant):
    """torque = sum r_i x (0, 0, k_f omega_i^2) with r_i from hummingbird.xacro:87,109,131,153: more thrust on the +y rotor rolls
    positive about x, more thrust on the -x rotor pitches positive about y; magnitude l k_f (omega_i^2 - omega_j^2) / I"""
    Ixx, Iyy, _ = plant.cfg.inertia
    om = np.array([400.0, 450.0, 400.0, 350.0])
    out = plant(init_state(om=om), const_cmd(om, 1))[0]
    tau = np.sum(np.cross(ROTOR_POS, np.c_[np.zeros(4), np.zeros(4), KF * om ** 2]), axis=0)
    assert tau[0] > 0 and abs(tau[1]) < 1e-12
    np.testing.assert_allclose(out[0, 10], DT * ARM * KF * (450.0 ** 2 - 350.0 ** 2) / Ixx, rtol=plant.tol * 10)
    np.testing.assert_allclose(out[0, 10], DT * tau[0] / Ixx, rtol=plant.tol * 10)
    assert out[0, 11] == 0.0
    om = np.array([350.0, 400.0, 450.0, 400.0])
    out = plant(init_state(om=om), const_cmd(om, 1))[0]
    np.testing.assert_allclose(out[0, 11], DT * ARM * KF * (450.0 ** 2 - 350.0 ** 2) / Iyy, rtol=plant.tol * 10)
    assert out[0, 11] > 0 and out[0, 10] == 0.0


def test_drag_torque_sign_follows_turning_direction(plant):
    """gazebo_motor_model.cpp:476-477: drag torque = -turning_direction * thrust * k_m about the rotor axis; hummingbird.xacro:72,94,
    116,138: rotors 0 and 2 turn cw (-1), rotors 1 and 3 ccw (+1) -> speeding up the cw pair yaws POSITIVE"""
    Izz = plant.cfg.inertia[2]
    om = np.array([450.0, 400.0, 450.0, 400.0])
    out = plant(init_state(om=om), const_cmd(om, 1))[0]
    tz = np.sum(-ROTOR_DIR * KF * om ** 2 * KM)
    assert tz > 0
    np.testing.assert_allclose(out[0, 12], DT * tz / Izz, rtol=plant.tol * 10)
    assert out[0, 10] == 0.0 and out[0, 11] == 0.0
    out = plant(init_state(om=om[::-1].copy()), const_cmd(om[::-1], 1))[0]
    np.testing.assert_allclose(out[0, 12], -DT * tz / Izz, rtol=plant.tol * 10)


def test_plant_wrench_agrees_with_the_reference_allocation_matrix(plant):
    """The controller side of the same physics is pinned by fixture G9: `AttitudeController.compute_allocation_matrix`
    (pkg/attitude_controller.py:94-104, imported by tests/golden/make_golden.py) maps omega^2 to (M_x, M_y, M_z, thrust).  The
    plant must produce exactly that wrench from the same rotor speeds, or controller and vehicle would disagree about signs."""
    from pathlib import Path
    A = np.load(Path(__file__).parent / "golden" / "g9_attitude.npz")["A"]
    I = np.asarray(plant.cfg.inertia)
    rng = np.random.default_rng(5)
    for _ in range(8):
        om = rng.uniform(250.0, 650.0, 4)
        out = plant(init_state(om=om), const_cmd(om, 1))[0]
        wrench = A @ om ** 2
        np.testing.assert_allclose(out[0, W] * I / DT, wrench[:3], rtol=plant.tol * 100, atol=1e-9 if plant.f64 else 2e-7)
        np.testing.assert_allclose((out[0, 5] / DT + G) * M, wrench[3], rtol=plant.tol * 200)


def test_rotor_drag_and_rolling_moment_for_a_lateral_velocity(plant):
    """gazebo_motor_model.cpp:453-469,484-489: per rotor force -|omega| c_d v_perp applied at the rotor (0.01 m above the centre of
    mass -> a pitching moment h F_x), rolling moment -|omega| c_r v_perp as a pure torque (parallel to the velocity!); a velocity
    along the rotor axis produces neither"""
    Ixx, Iyy, _ = plant.cfg.inertia
    vx = 1.5
    out = plant(init_state(v=(vx, 0, 0)), const_cmd([HOVER] * 4, 1))[0]
    S = 4 * HOVER
    Fx = -S * C_DRAG * vx
    np.testing.assert_allclose(out[0, 3] - vx, DT * Fx / M, rtol=1e-3 if not plant.f64 else 1e-9)
    np.testing.assert_allclose(out[0, 11], DT * (ROTOR_Z * Fx) / Iyy, rtol=plant.tol * 10)
    np.testing.assert_allclose(out[0, 10], DT * (-S * C_ROLL * vx) / Ixx, rtol=plant.tol * 10)
    assert out[0, 12] == 0.0 and out[0, 4] == 0.0
    out = plant(init_state(v=(0, 0, 1.0)), const_cmd([HOVER] * 4, 1))[0]
    assert out[0, 3] == 0.0 and out[0, 4] == 0.0 and np.all(out[0, W] == 0.0)


def test_drag_uses_the_velocity_component_in_the_rotor_plane_of_a_tilted_vehicle(plant):
    """world-frame closed form for a vehicle pitched by 20 deg flying along world x: thrust along n = R e_z, drag -S c_d (v - (v.n) n)"""
    th = np.deg2rad(20.0)
    q = (np.cos(th / 2), 0.0, np.sin(th / 2), 0.0)
    n = np.array([np.sin(th), 0.0, np.cos(th)])
    v = np.array([2.0, 0.5, -0.3])
    out = plant(init_state(v=v, q=q), const_cmd([HOVER] * 4, 1))[0]
    F = KF * 4 * HOVER ** 2 * n - 4 * HOVER * C_DRAG * (v - v.dot(n) * n)
    acc = F / M - np.array([0, 0, G])
    np.testing.assert_allclose(out[0, V] - v, DT * acc, rtol=2e-3 if not plant.f64 else 1e-9, atol=1e-7 if not plant.f64 else 0)


def test_free_body_rotation(plant):
    """no rotors: a pure spin about body z turns the attitude by omega_z t (quaternion stays unit); a symmetric body (I_xx = I_yy)
    precesses: (w_x, w_y) turn at Om = (I_zz - I_xx) / I_xx w_z with w_z constant (Euler's equations I w' = -w x I w).  The kernel
    integrates the rates with explicit Euler, whose closed form for this linear system is (w_x + i w_y)_k = (1 + i Om dt)^k —
    the exact rotation e^{i Om t} to first order in dt"""
    n = 250
    out = plant(init_state(w=(0, 0, 2.0), om=(0,) * 4), const_cmd([0] * 4, n))[0]
    psi = 2.0 * DT * np.arange(1, n + 1)
    np.testing.assert_allclose(out[:, 6], np.cos(psi / 2), atol=2e-6)
    np.testing.assert_allclose(out[:, 9], np.sin(psi / 2), atol=2e-6)
    np.testing.assert_allclose(np.sum(out[:, Q] ** 2, axis=1), 1.0, atol=1e-6)
    assert np.all(out[:, 12] == 2.0)
    Ixx, _, Izz = plant.cfg.inertia
    out = plant(init_state(w=(1.0, 0, 5.0), om=(0,) * 4), const_cmd([0] * 4, n))[0]
    Om = (Izz - Ixx) / Ixx * 5.0
    t = DT * np.arange(1, n + 1)
    z = (1 + 1j * Om * DT) ** np.arange(1, n + 1)  # w_x' = -Om w_y, w_y' = Om w_x, one explicit Euler step per tick
    np.testing.assert_allclose(out[:, 10], z.real, atol=1e-9 if plant.f64 else 2e-5)
    np.testing.assert_allclose(out[:, 11], z.imag, atol=1e-9 if plant.f64 else 2e-5)
    np.testing.assert_allclose(np.abs(z - np.exp(1j * Om * t)).max(), 0.5 * Om ** 2 * DT * t[-1], rtol=0.05)  # and that IS the rotation, to O(dt)
    np.testing.assert_allclose(out[:, 12], 5.0, rtol=1e-6)


def test_contact_latch_geometry_and_platform_extrapolation(plant):
    """contact <=> drone base bottom (z - 0.06, hummingbird.xacro:31) at or below the landing surface (0.455 m) inside the 1 x 1 m
    deck widened by half the base (0.05 m); the flag latches; between manager ticks the platform moves on with its velocity"""
    def run(dx, dy, z, n=3, u=0.0):
        return plant(init_state(p=(dx, dy, z), mp=(0.0, 0.0, u, 0.0)), const_cmd([HOVER] * 4, n))[0]
    zc = PLATFORM_TOP + BOTTOM
    assert run(0.0, 0.0, zc - 1e-3)[0, CONTACT] == 1.0
    assert run(0.0, 0.0, zc + 1e-3)[:, CONTACT].max() == 0.0
    assert run(HALF - 1e-3, 0.0, zc - 1e-3)[0, CONTACT] == 1.0
    assert run(HALF + 1e-3, 0.0, zc - 1e-3)[:, CONTACT].max() == 0.0
    assert run(0.0, -(HALF - 1e-3), zc - 1e-3)[0, CONTACT] == 1.0
    assert run(0.0, -(HALF + 1e-3), zc - 1e-3)[:, CONTACT].max() == 0.0
    assert run(3.0, 0.0, 0.1)[:, CONTACT].max() == 0.0          # below the deck height but beside it: the ground is not the platform
    o = run(0.0, 0.0, zc - 1e-3, n=400, u=2.0)                  # the platform drives away from under the drone: the latch keeps the contact
    np.testing.assert_allclose(o[:, MPX], 2.0 * DT * np.arange(1, 401), rtol=plant.tol * 400)
    assert o[-1, MPX] > HALF and np.all(o[:, CONTACT] == 1.0)
