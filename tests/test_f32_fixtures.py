"""float32 legs of the reference-fixture tests (What's weak #1): the oracle's float32 build — the restatement the float32
kernel is held to bit for bit, and which is rewritten in lock-step with it whenever a form changes — against the golden vectors the
reference's own Python produced.  The same checks run against the HIP library in tests/test_gpu_operators.py (-m gpu).  Bounds and their
derivations: tests/fixture_checks.py."""
import pytest

import fixture_checks as fc
from dql_multirotor_landing_amd.config import F32, F64


@pytest.fixture(scope="module")
def be():
    return fc.OracleBackend()


@pytest.mark.parametrize("dtype", [F64, F32])
def test_g8_butterworth(be, dtype):
    """pkg/filters.py:98-109; float32 = transposed direct form with three states (csrc/dql_device.hpp butterworth)"""
    fc.check_g8_butterworth(be, dtype)


@pytest.mark.parametrize("dtype", [F64, F32])
def test_g8_kalman(be, dtype):
    """pkg/filters.py:19-80, R = 0 and R = 0.01"""
    fc.check_g8_kalman(be, dtype)


@pytest.mark.parametrize("dtype", [F64, F32])
def test_g8_pid(be, dtype):
    """pkg/pid.py:62-104: integral, windup clip, Butterworth on the error, output clip (med3 in float32)"""
    fc.check_g8_pid(be, dtype)


@pytest.mark.parametrize("dtype", [F64, F32])
def test_g9_rotor_speeds(be, dtype):
    """pkg/attitude_controller.py:107-156 down to the commanded rotor speeds, clamp at zero included"""
    fc.check_g9_rotor_speeds(be, dtype)


def test_g9_x_axis_closed_form(be):
    fc.check_g9_xonly_form(be)


@pytest.mark.parametrize("dtype,carry", [(F64, 0), (F32, 0), (F32, 4), (F32, 5)])
def test_g11_platform(be, dtype, carry):
    """pkg/moving_platform.py:87-127; carry = 4 / 5: sine and cosine carried by rotation through an agent period's manager ticks (float32 step)"""
    fc.check_g11_platform(be, dtype, carry)


def test_g12_manager_tick_f32(be):
    """scripts/manager_node.py:192-214 + pkg/observation_utils.py:99-158 in float32, all three series of the fixture"""
    fc.check_g12_manager_f32(be)


def test_angle_bin_from_tangent_equals_argmin_of_the_euler_angle():
    """Round 5: the float32 fused step picks the angle bin of the discrete state from three comparisons of tan^2 against the grid's bin boundaries
    (csrc/dql_device.hpp angle_bin_from_tangent) instead of atan2 + argmin |grid - clip(angle)| (pkg/mdp.py:318-324).  Held against the reference-shaped
    form: the index the step stored == discretise() — the G1-pinned operator — applied to the step's own latched observation and the Euler angle
    recomputed from its quaternion in float64, for every env whose angle is not within 2e-6 rad of a bin boundary (float32 rounding of the rotation
    matrix entries), x-axis and two-axis, pitching hard enough to reach the outer bins."""
    import numpy as np
    from dql_multirotor_landing_amd.config import DqlConfig
    from oracle import oracle as orc
    for kw in (dict(), dict(two_axis=1)):
        cfg = DqlConfig(dtype=F32, **kw)
        o = orc.Oracle(cfg, 1536, seed=13, n_threads=4)
        names, inames = o.field_names(), o.field_names(True)
        step = 2 * cfg.theta_max / 6.0
        bounds = (np.arange(6) - 2.5) * step
        seen = set()
        for rounds in range(12):
            o.train_steps(5, 1.0)
            reals, ints = o.get_fields()
            g = lambda k: reals[names.index(k)]
            w, x, y, z = g("qw"), g("qx"), g("qy"), g("qz")
            R20, R21, R22 = 2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)
            R00, R10 = 1 - 2 * (y * y + z * z), 2 * (x * y + w * z)
            pitch = np.arctan2(-R20, np.sqrt(R00 * R00 + R10 * R10)); roll = np.arctan2(R21, R22)
            for ax, ang, obs in ((0, pitch, ("obs_p_x", "obs_v_x", "obs_a_x")),) + (((1, -roll, ("obs_p_y", "obs_v_y", "obs_a_y")),) if kw else ()):
                want = orc.discretise(cfg, g(obs[0]), g(obs[1]), g(obs[2]), ang)
                got = ints[ax]
                clear = np.abs(ang[:, None] - bounds[None, :]).min(axis=1) > 2e-6
                ok = (want >= 0) & clear
                assert ok.mean() > 0.99
                np.testing.assert_array_equal(got[ok], want[ok])
                seen |= set((got[ok] % 7).tolist())
        assert len(seen) >= 5, seen   # the random policy pitches through most of the grid
