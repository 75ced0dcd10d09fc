"""Stateless C-ABI operators on the GPU against the golden vectors of the reference (bit-exact in float64)."""
import numpy as np
import pytest

from dql_multirotor_landing_amd.config import DqlConfig, F64, N_CELLS

pytestmark = pytest.mark.gpu


def unpack(idx):
    idx = np.asarray(idx)
    return np.stack([idx // 189, (idx // 63) % 3, (idx // 21) % 3, (idx // 7) % 3, idx % 7], axis=-1)


def pack(s):
    s = np.asarray(s).astype(np.int64)
    return ((((s[..., 0] * 3 + s[..., 1]) * 3 + s[..., 2]) * 3 + s[..., 3]) * 7 + s[..., 4]).astype(np.int32)


@pytest.fixture(scope="module")
def ops():
    from dql_multirotor_landing_amd import ops
    return ops


@pytest.mark.parametrize("level", range(5))
def test_discretise_golden(ops, golden_dir, level):
    g = np.load(golden_dir / "g1_discretise.npz")
    x = g[f"in_{level}"]
    idx = ops.discretise(DqlConfig(working_curriculum_step=level, dtype=F64), x[:, 0], x[:, 1], x[:, 2], x[:, 3])
    np.testing.assert_array_equal(unpack(idx), g[f"state_{level}"])


@pytest.mark.parametrize("level", [0, 3])
def test_mdp_trace_golden(ops, golden_dir, level):
    t = np.load(golden_dir / "g2_traces.npz")[f"trace_{level}"]
    cfg = DqlConfig(working_curriculum_step=level, dtype=F64)
    ms = np.zeros((8, 1)); ms[7] = 8
    prev = np.array([-1], dtype=np.int32)
    for row in t[:700]:
        op, act = int(row[0]), int(row[1])
        obs = np.array([row[2], row[3], row[4], row[5], row[6], row[7], row[8]]).reshape(7, 1)
        if op == 0:
            ms[0] = 0.0; ms[4] = 0.0; ms[5] = 0; ms[6] = 0; ms[7] = 8
            prev = ops.discretise(cfg, obs[0], obs[2], obs[3], obs[4]).astype(np.int32)
            continue
        ms, idx, rew, done = ops.mdp_transition(cfg, [act], obs, ms, prev)
        np.testing.assert_array_equal(unpack(idx)[0], row[9:14].astype(int))
        assert int(ms[7, 0]) == int(row[14]) and rew[0] == row[15] and int(done[0]) == int(row[16])
        assert ms[0, 0] == row[17] and ms[4, 0] == row[18]
        prev = idx.astype(np.int32)


def test_agent_golden(ops, golden_dir):
    g = np.load(golden_dir / "g4_agent.npz")
    qa = np.zeros(N_CELLS); qb = np.zeros(N_CELLS); cnt = np.zeros(N_CELLS)
    sa = g["upd_sa"]
    ops.agent_update(qa, qb, cnt, pack(sa[:, :5]) * 3 + sa[:, 5], pack(g["upd_ns"]), g["upd_alpha"], 0.99, g["upd_reward"])
    np.testing.assert_array_equal(qa, g["upd_Qa"].ravel())
    np.testing.assert_array_equal(cnt, g["upd_count"].ravel())
    assert not qb.any()
    A = np.load(golden_dir / "assets" / "Q_table_a.npy"); B = np.load(golden_dir / "assets" / "Q_table_b.npy")
    np.testing.assert_array_equal(ops.agent_predict(A, B, np.arange(945)), g["predict_actions"])
