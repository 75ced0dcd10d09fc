"""ctypes loader of csrc/libdql_hip.so (C ABI: include/dql.h).  There is no CPU fallback: if the library is
missing, or no MI355X is visible when a device call is made, this raises."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

from .config import DqlConfigC, N_CHECK_CODES

CSRC = Path(__file__).resolve().parent / "csrc"
LIB_PATH = Path(os.environ.get("DQL_LIB_PATH", CSRC / "libdql_hip.so"))  # override: A/B builds of the kernel

OK, EINVAL, EHIP, ESTATE, ENOMEM, ERCCL, EPEER = 0, -1, -2, -3, -4, -5, -6
ABI_VERSION = 6
COMM_ID_BYTES = 128
P2P_HANDLE_BYTES = 64
P2P_MAX_RANKS = 8
OP_SUM, OP_MAX = 0, 1


class DqlStatsC(C.Structure):
    _fields_ = [("agent_steps", C.c_int64), ("decisions", C.c_int64), ("episodes", C.c_int64),
                ("by_code", C.c_int64 * N_CHECK_CODES), ("reward_sum", C.c_double), ("physics_ticks", C.c_int64)]


# every symbol include/dql.h and include/dql_diag.h (dql_diag_*: measurement and self-test, not part of the drop-in boundary) declare: name -> (restype, argtypes)
_vp, _i32, _i64, _u64, _dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_double
_cfgp = C.POINTER(DqlConfigC)
SYMBOLS = {
    "dql_abi_version": (C.c_int, []),
    "dql_last_error": (C.c_char_p, []),
    "dql_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "dql_config_default": (C.c_int, [_cfgp]),
    "dql_create": (C.c_int, [_cfgp, C.c_int, _i64, _u64, _i64, C.POINTER(_vp)]),
    "dql_destroy": (C.c_int, [_vp]),
    "dql_sync": (C.c_int, [_vp]),
    "dql_n_envs": (C.c_int, [_vp, C.POINTER(_i64)]),
    "dql_state_bytes_per_env": (C.c_int, [_vp, C.POINTER(_i64)]),
    "dql_set_alpha_table": (C.c_int, [_vp, _vp, _i32]),
    "dql_set_curriculum": (C.c_int, [_vp, _i32]),
    "dql_reset": (C.c_int, [_vp, _vp]),
    "dql_step": (C.c_int, [_vp, _vp]),
    "dql_step_dev": (C.c_int, [_vp, _vp]),
    "dql_step_outputs": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dql_train_steps": (C.c_int, [_vp, _i32, _dbl]),
    "dql_eval_steps": (C.c_int, [_vp, _i32]),
    "dql_get_states": (C.c_int, [_vp, _vp, _vp]),
    "dql_get_rewards": (C.c_int, [_vp, _vp]),
    "dql_get_dones": (C.c_int, [_vp, _vp, _vp]),
    "dql_get_actions": (C.c_int, [_vp, _vp]),
    "dql_get_sim_state": (C.c_int, [_vp, _vp, _i32]),
    "dql_set_sim_state": (C.c_int, [_vp, _vp, _i32]),
    "dql_get_sim_ints": (C.c_int, [_vp, _vp, _i32]),
    "dql_set_sim_ints": (C.c_int, [_vp, _vp, _i32]),
    "dql_n_fields": (C.c_int, [C.POINTER(_i32), C.POINTER(_i32)]),
    "dql_field_name": (C.c_char_p, [_i32, _i32]),
    "dql_get_obs": (C.c_int, [_vp, _vp]),
    "dql_get_tables": (C.c_int, [_vp, _vp, _vp, _vp]),
    "dql_set_tables": (C.c_int, [_vp, _vp, _vp, _vp]),
    "dql_transfer": (C.c_int, [_vp, _i32, _dbl]),
    "dql_set_sync_period": (C.c_int, [_vp, _i32]),
    "dql_set_windowed": (C.c_int, [_vp, _i32]),
    "dql_diag_accum_dev_ptr": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_i64)]),
    "dql_set_window_buffer": (C.c_int, [_vp, _vp]),
    "dql_stream_handle": (C.c_int, [_vp, C.POINTER(_vp)]),
    "dql_flush": (C.c_int, [_vp]),
    "dql_apply_accum": (C.c_int, [_vp]),
    "dql_get_accum": (C.c_int, [_vp, _vp]),
    "dql_set_accum": (C.c_int, [_vp, _vp]),
    "dql_get_step_index": (C.c_int, [_vp, C.POINTER(_i64)]),
    "dql_set_step_index": (C.c_int, [_vp, _i64]),
    "dql_publish_tables": (C.c_int, [_vp]),
    "dql_comm_unique_id": (C.c_int, [_vp]),
    "dql_comm_create": (C.c_int, [C.c_int, _i32, _i32, _vp, C.POINTER(_vp)]),
    "dql_comm_destroy": (C.c_int, [_vp]),
    "dql_comm_info": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "dql_comm_allreduce_f64": (C.c_int, [_vp, _vp, _i64, _i32]),
    "dql_comm_allreduce_i64": (C.c_int, [_vp, _vp, _i64, _i32]),
    "dql_comm_allgather_u64": (C.c_int, [_vp, _vp, _i64, _vp]),
    "dql_comm_barrier": (C.c_int, [_vp]),
    "dql_attach_comm": (C.c_int, [_vp, _vp]),
    "dql_allreduce_window": (C.c_int, [_vp]),
    "dql_p2p_create": (C.c_int, [_vp, C.c_int32, C.c_int32, _vp]),
    "dql_p2p_connect": (C.c_int, [_vp, _vp]),
    "dql_p2p_connect_local": (C.c_int, [_vp, _vp]),
    "dql_p2p_exchange_window": (C.c_int, [_vp]),
    "dql_p2p_push_window": (C.c_int, [_vp]),
    "dql_p2p_wait_window": (C.c_int, [_vp]),
    "dql_p2p_status": (C.c_int, [_vp, C.POINTER(C.c_int32)]),
    "dql_diag_sync_time_ms": (C.c_int, [_vp, C.POINTER(_dbl), C.POINTER(_i64)]),
    "dql_stats_get": (C.c_int, [_vp, C.POINTER(DqlStatsC)]),
    "dql_stats_reset": (C.c_int, [_vp]),
    "dql_diag_timer_start": (C.c_int, [_vp]),
    "dql_diag_timer_stop": (C.c_int, [_vp, C.POINTER(_dbl)]),
    "dql_diag_kernel_timer": (C.c_int, [_vp, _i32]),
    "dql_diag_kernel_time_ms": (C.c_int, [_vp, C.POINTER(_dbl), C.POINTER(_i64)]),
    "dql_diag_delay": (C.c_int, [_vp, _dbl]),
    "dql_set_option": (C.c_int, [_vp, C.c_char_p, _i32]),
    "dql_episode_log_enable": (C.c_int, [_vp, _i32]),
    "dql_episode_log_read": (C.c_int, [_vp, _vp, _vp, _i32, C.POINTER(_i32)]),
    "dql_episode_log_read_words": (C.c_int, [_vp, _vp, _vp, _i32, _i32, C.POINTER(_i32)]),
    "dql_discretise": (C.c_int, [_cfgp, C.c_int, _vp, _vp, _vp, _vp, _i64, _vp]),
    "dql_mdp_transition": (C.c_int, [_cfgp, C.c_int, _i64, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "dql_manager_run": (C.c_int, [_cfgp, C.c_int, _i64, _i64, _vp, _vp, _u64, _vp]),
    "dql_plant_run": (C.c_int, [_cfgp, C.c_int, _i64, _i64, _vp, _vp, _vp]),
    "dql_butterworth_run": (C.c_int, [_cfgp, C.c_int, _vp, _i64, _vp]),
    "dql_kalman_run": (C.c_int, [_cfgp, C.c_int, _vp, _vp, _i64, _vp]),
    "dql_pid_run": (C.c_int, [_cfgp, C.c_int, _vp, _vp, _i64, _vp, _vp]),
    "dql_attitude_run": (C.c_int, [_cfgp, C.c_int, _vp, _vp, _vp, _i64, _i32, _vp]),
    "dql_platform_run": (C.c_int, [_cfgp, C.c_int, _i64, _i32, _vp]),
    "dql_diag_selftest_sqrt": (C.c_int, [C.c_int, C.c_uint32, C.c_uint32, C.POINTER(_i64)]),
    "dql_place": (C.c_int, [_cfgp, C.c_int, _vp, _vp, _i64, _vp]),
    "dql_agent_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "dql_agent_destroy": (C.c_int, [_vp]),
    "dql_agent_set_tables": (C.c_int, [_vp, _vp, _vp, _vp]),
    "dql_agent_get_tables": (C.c_int, [_vp, _vp, _vp, _vp]),
    "dql_agent_predict_resident": (C.c_int, [_vp, _vp, _i64, _vp]),
    "dql_agent_update_resident": (C.c_int, [_vp, _vp, _vp, _vp, _dbl, _vp, _i64, C.c_uint32, _vp, _vp, _vp, _vp, _vp]),
    "dql_agent_mirror_predict": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int32, C.c_int32, _vp]),
    "dql_agent_mirror_update": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int32, C.c_int32, C.c_int32, _dbl, _dbl, _dbl, C.c_uint32, C.c_int32, C.c_int32]),
    "dql_agent_mirror_update_deferred": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int32, C.c_int32, C.c_int32, _dbl, _dbl, _dbl, C.c_uint32, C.c_int32, C.c_int32]),
    "dql_agent_mirror_complete": (C.c_int, [_vp]),
    "dql_agent_transfer": (C.c_int, [C.c_int, _vp, _vp, _i32, _dbl]),
    "dql_agent_predict": (C.c_int, [C.c_int, _vp, _vp, _vp, _i64, _vp]),
    "dql_agent_update": (C.c_int, [C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _dbl, _vp, _i64, C.c_uint32, _vp, _vp]),
}

_lib = None


def load():
    """Load the HIP library; raise if it has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise RuntimeError(f"{LIB_PATH} not found: build the HIP extension first (__graft_entry__.build()); there is no CPU fallback")
        lib = C.CDLL(str(LIB_PATH))
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the library does not export what the header declares
            fn.restype = res
            fn.argtypes = args
        if lib.dql_abi_version() != ABI_VERSION:
            raise RuntimeError("libdql_hip.so ABI version mismatch")
        _lib = lib
    return _lib


def check(rc: int):
    """Map C status codes to the exception types the reference raises for the same conditions."""
    if rc == OK:
        return
    msg = load().dql_last_error().decode(errors="replace")
    if rc in (EINVAL, ESTATE):
        raise ValueError(msg)
    if rc == ENOMEM:
        raise MemoryError(msg)
    raise RuntimeError(msg)
