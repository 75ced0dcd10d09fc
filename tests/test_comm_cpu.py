"""CPU-side checks of the multi-rank plumbing that needs no GPU: how ranks find rank 0's unique id, and that
`bench.py --gpus N` never degrades to a silent single-GPU measurement."""
import ctypes as C
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent

_PRINT_PATH = "import sys; sys.path.insert(0, %r); from dql_multirotor_landing_amd.comm import id_file_path; print(id_file_path(0)); print(id_file_path(1))" % str(ROOT)


def _child_paths(port, extra=None):
    env = dict(os.environ, MASTER_PORT=str(port))
    env.pop("DQL_COMM_ID_FILE", None)
    env.update(extra or {})
    return subprocess.run([sys.executable, "-c", _PRINT_PATH], env=env, capture_output=True, text=True, check=True).stdout.split("\n")[:2]


def test_sibling_ranks_agree_on_the_id_file():
    """Ranks are siblings (children of the launcher): same parent => same file name, without any communication; another job
    (other port, or another parent) gets another name, so a stale file is never picked up."""
    a, b = _child_paths(29500), _child_paths(29500)
    assert a == b and a[0] != a[1]
    assert _child_paths(29501)[0] != a[0]
    assert str(os.getpid()) in a[0]  # keyed by the parent (this process) ...
    via_shell = subprocess.run(["bash", "-c", f"exec 3>&1; ( {sys.executable} -c {_PRINT_PATH!r} ) "], env=dict(os.environ, MASTER_PORT="29500"),
                               capture_output=True, text=True, check=True).stdout.split("\n")[0]
    assert via_shell != a[0]  # ... and another parent gives another name
    assert _child_paths(29500, {"DQL_COMM_ID_FILE": "/tmp/x.id"}) == ["/tmp/x.id", "/tmp/x.id.1"]


def test_comm_entry_points_fail_loudly_without_gpu():
    from dql_multirotor_landing_amd import _lib
    lib = _lib.load()
    n = C.c_int(0)
    if lib.dql_device_count(C.byref(n)) == 0 and n.value > 0:
        pytest.skip("a GPU is visible here")
    h = C.c_void_p()
    uid = (C.c_uint8 * _lib.COMM_ID_BYTES)()
    assert lib.dql_comm_create(0, 0, 1, uid, C.byref(h)) == _lib.EHIP and not h.value
    assert lib.dql_comm_create(0, 2, 2, uid, C.byref(h)) == _lib.EINVAL  # rank out of range
    assert lib.dql_comm_barrier(None) == _lib.EINVAL
    assert lib.dql_comm_destroy(None) == 0
    from dql_multirotor_landing_amd.comm import RcclComm
    with pytest.raises(RuntimeError):
        RcclComm(0, 1, 0)
    assert RcclComm.from_env() is None or os.environ.get("WORLD_SIZE", "1") != "1"


def test_bench_gpus_2_without_two_gpus_fails_loudly():
    """The driver starts `python bench.py --gpus N` without a launcher: it must start N ranks or fail, never print n_gpus: 1."""
    from dql_multirotor_landing_amd import _lib
    n = C.c_int(0)
    if _lib.load().dql_device_count(C.byref(n)) == 0 and n.value >= 2:
        pytest.skip("two GPUs are visible here")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-curriculum",
                        "--large-envs", "0"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert '"n_gpus"' not in r.stdout
    assert "fewer than 2 ranks ran" in r.stderr and "GPU" in r.stderr
    # a launcher that started a different number of ranks than --gpus says is an error too
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "4", "--steps", "3", "--no-cpu-baseline", "--no-curriculum"], cwd=ROOT,
                       env=dict(env, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29512"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 2 and "launcher started 2" in r.stderr


def test_gather_via_files_two_ranks(tmp_path, monkeypatch):
    """comm.gather_via_files: the bootstrap all-gather of the peer-to-peer exchange's IPC handles (no collective library involved):
    payloads come back in rank order on every rank, and the data files are gone afterwards."""
    import threading
    from dql_multirotor_landing_amd import comm
    monkeypatch.setenv("DQL_COMM_ID_FILE", str(tmp_path / "boot.id"))
    res = {}

    def run(rank):
        res[rank] = comm.gather_via_files("t0", rank, 2, bytes([rank]) * 64, timeout_s=30)

    th = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(60)
    assert res[0] == res[1] == [bytes([0]) * 64, bytes([1]) * 64]
    assert not list(tmp_path.glob("*.bin"))


def test_bootstrap_files_of_a_killed_job_are_polled_past(tmp_path, monkeypatch):
    """with an explicit DQL_COMM_ID_FILE the name carries no job nonce, so the header does: a hash of DQL_COMM_JOB_ID, or of
    TORCHELASTIC_RUN_ID + the parent's (pid, start time).  Leftovers of a killed job — a file of another job (other nonce, MASTER_PORT or world
    size), however recent, a stray `.done` marker — must not be taken for this job's; and a valid file stays valid however OLD it is (a rank
    that starts minutes after rank 0 wrote it)."""
    import threading
    import time
    from dql_multirotor_landing_amd import comm
    monkeypatch.setenv("DQL_COMM_ID_FILE", str(tmp_path / "boot.id"))
    monkeypatch.setenv("MASTER_PORT", "29533")
    monkeypatch.setenv("DQL_COMM_JOB_ID", "job-A")
    f = tmp_path / "stale.bin"
    f.write_bytes(comm._header(2) + b"x" * 64)
    assert comm._read_fresh(f, 2, 64) == b"x" * 64
    assert comm._read_fresh(f, 3, 64) is None                     # another world size
    old = time.time() - 3600
    os.utime(f, (old, old))
    assert comm._read_fresh(f, 2, 64) == b"x" * 64                # an hour old and still this job's: a late rank is not locked out
    monkeypatch.setenv("DQL_COMM_JOB_ID", "job-B")
    assert comm._read_fresh(f, 2, 64) is None                     # the same file seen by ANOTHER job (relaunch on the same port and world size)
    f.write_bytes(comm._header(2) + b"x" * 64)                    # ... written by job B a moment ago
    monkeypatch.setenv("DQL_COMM_JOB_ID", "job-A")
    assert comm._read_fresh(f, 2, 64) is None                     # fresh, same port, same world: still not job A's
    monkeypatch.delenv("DQL_COMM_JOB_ID")
    h1 = comm._header(2)
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "run-7")
    assert comm._header(2) != h1 and len(h1) == 24               # without a job id: the run id and the parent's (pid, start time) make the nonce
    h_run7 = comm._header(2)
    monkeypatch.setenv("TORCHELASTIC_RESTART_COUNT", "1")         # attempt k + 1 of an elastic job (same agent, run id and port) is another job
    assert comm._header(2) not in (h1, h_run7)
    monkeypatch.setattr(comm, "_parent_token", lambda: "1_1")
    assert comm._header(2) != h1
    monkeypatch.undo()
    monkeypatch.setenv("DQL_COMM_ID_FILE", str(tmp_path / "boot.id"))
    monkeypatch.setenv("MASTER_PORT", "29534")
    f.write_bytes(comm._header(2) + b"x" * 64)
    monkeypatch.setenv("MASTER_PORT", "29533")
    assert comm._read_fresh(f, 2, 64) is None                     # another job's port
    f.write_bytes(b"x" * 64)
    assert comm._read_fresh(f, 2, 64) is None                     # no header at all (what round 2 wrote)
    old_header = b"DQLC" + (29533).to_bytes(4, "little") + (2).to_bytes(4, "little") + b"\0\0\0\0"
    # a full gather with leftovers of rank 1 in place: rank 0 must wait for the REAL rank 1
    base = comm.id_file_path(0)
    stale_bin = base.with_name(f"{base.stem}.t9.1.bin"); stale_done = base.with_name(f"{base.stem}.t9.1.done")
    stale_bin.write_bytes(old_header + b"S" * 64 + b"\0" * 8); stale_done.write_bytes(old_header + b"\0" * 8)   # round 3's header: another "job"
    res = {}

    def run(rank, delay):
        time.sleep(delay)
        res[rank] = comm.gather_via_files("t9", rank, 2, bytes([rank + 1]) * 64, timeout_s=30)

    th = [threading.Thread(target=run, args=(0, 0.0)), threading.Thread(target=run, args=(1, 0.5))]
    for t in th:
        t.start()
    for t in th:
        t.join(60)
    assert res[0] == res[1] == [bytes([1]) * 64, bytes([2]) * 64]
