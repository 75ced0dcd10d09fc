"""CPU-side checks of the boundary: the library loads, exports every symbol include/dql.h declares, the ctypes
mirror of dql_config matches the C layout, and the product fails loudly (no CPU fallback) without a GPU."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build_hip()
    from dql_multirotor_landing_amd import _lib
    return _lib.load()


def test_every_declared_symbol_is_exported_and_bound(lib):
    from dql_multirotor_landing_amd import _lib
    hdr = (ROOT / "include" / "dql.h").read_text()
    diag = (ROOT / "include" / "dql_diag.h").read_text()
    declared = set(re.findall(r"\b(dql_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"dql_config", "dql_stats", "dql_ctx", "dql_status"}
    declared_diag = set(re.findall(r"\b(dql_[a-z0-9_]+)\s*\(", diag)) - {"dql_last_error"}
    assert declared and declared_diag, "no declarations found"
    # the drop-in surface and the lab equipment are two headers; everything measurement-only carries the dql_diag_ prefix
    assert all(n.startswith("dql_diag_") for n in declared_diag) and not any(n.startswith("dql_diag_") for n in declared)
    assert declared | declared_diag == set(_lib.SYMBOLS), f"headers vs binding: {(declared | declared_diag) ^ set(_lib.SYMBOLS)}"
    for name in declared | declared_diag:
        assert getattr(lib, name) is not None
    assert lib.dql_abi_version() == _lib.ABI_VERSION == 6
    # the host classes that mirror the reference (mdp, agent, env, trainer) never touch a diagnostic symbol
    for f in ("mdp.py", "double_q_learning.py", "landing_simulation_env.py", "trainer.py", "promotion.py", "dist.py", "comm.py"):
        assert "dql_diag_" not in (ROOT / "dql_multirotor_landing_amd" / f).read_text(), f


def test_config_layout_and_defaults_match_c(lib):
    from dql_multirotor_landing_amd.config import DqlConfig, DqlConfigC
    c = DqlConfigC()
    assert lib.dql_config_default(C.byref(c)) == 0
    py = DqlConfig().to_c()
    for name, _ in DqlConfigC._fields_:
        a, b = getattr(c, name), getattr(py, name)
        if hasattr(a, "__len__"):
            assert list(a) == list(b), name
        else:
            assert a == b, name
    assert bytes(c) == bytes(py)  # identical bytes => identical layout incl. padding


def test_field_names(lib):
    nr, ni = C.c_int32(), C.c_int32()
    lib.dql_n_fields(C.byref(nr), C.byref(ni))
    assert (nr.value, ni.value) == (64, 7)
    from oracle.oracle import Oracle
    from dql_multirotor_landing_amd.config import DqlConfig
    o = Oracle(DqlConfig(), 1)
    assert [lib.dql_field_name(i, 0).decode() for i in range(64)] == o.field_names()
    assert [lib.dql_field_name(i, 1).decode() for i in range(7)] == o.field_names(True)


def test_no_cpu_fallback_without_gpu(lib):
    """On a box without a GPU every device entry point must fail loudly instead of computing on the CPU."""
    n = C.c_int(0)
    rc = lib.dql_device_count(C.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is visible here")
    from dql_multirotor_landing_amd.config import DqlConfig
    from dql_multirotor_landing_amd.engine import Engine
    from dql_multirotor_landing_amd import ops
    with pytest.raises(RuntimeError):
        Engine(DqlConfig(), 8)
    with pytest.raises(RuntimeError):
        ops.discretise(DqlConfig(), np.zeros(2), np.zeros(2), np.zeros(2), np.zeros(2))


def test_product_never_imports_oracle():
    for p in (ROOT / "dql_multirotor_landing_amd").rglob("*.py"):
        assert "oracle" not in p.read_text().lower().replace("the oracle", "").replace("cpu oracle", ""), p
