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
