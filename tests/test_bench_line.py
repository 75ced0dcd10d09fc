"""bench.py prints ONE line the driver can parse: compact (<= 7 KB whatever the number of curriculum seeds or ranks), with the
contract's keys, `roofline` and `cpu_baseline`; the per-seed detail goes to a side file.  Round 3's line was 26 KB and the driver, which keeps
the last 8 KB of stdout, recorded `parsed: null`.  CPU only: the records are synthetic, the relay test starts stub children."""
import json
import subprocess
import sys
from pathlib import Path
from types import SimpleNamespace

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def synthetic_record(n_seeds=12, n_gpus=1):
    """the full record bench.main() assembles, filled with plausible values and deliberately long strings"""
    roof = {"bound": "valu_issue", "frac_is": "hbm_algorithmic", "hbm_real_frac": 0.0198, "valu_issue_frac": 0.55,
            "achieved": 1712.35, "peak": 8000.0, "unit": "GB/s", "frac": 0.214, "traffic": 59997858.1, "traffic_note": "x" * 400,
            "kernel": "k_step", "kernel_avg_ms": 0.3947, "kernel_launches_timed": 125, "agent_periods_per_launch": 16, "algorithmic_bytes_per_env_step": 328,
            "env_steps_per_launch": 2060893.6, "note": "y" * 300, "kernel_avg_ms_event_pairs": 0.41}
    issue = {"valu_instr_per_env_wave_per_period": 8302.5, "env_waves_per_simd": 2.0, "cycles_per_instr": 2, "frac_at_2p4_ghz": 0.55, "measured_clock_ghz": 1.9,
             "frac_at_measured_clock": 0.70, "source": "z" * 200}
    block = lambda envs: {"workload": "w" * 300, "envs": envs, "value": 2.3e8, "unit": "env-steps/s", "steps": 2000, "warmup": 200, "ms_per_step": 0.017,
                          "device_ms_per_step": 0.017, "periods_per_launch": 16, "roofline": dict(roof), "valu_issue": dict(issue)}
    levels = [{"level": k, "promoted": True, "exhausted": False, "episodes": 1234567, "agent_periods": 4096, "wall_s": 0.41234567,
               "population_success_at_promotion": 0.95123456, "online_success_rate_at_handover": 0.95123456} for k in range(5)]
    runs = [{"seed": s, "wall_to_stage4_s": 1.5, "wall_all_levels_s": 2.3, "promoted_levels": 5, "levels": levels,
             "stage4_greedy_4096_episodes": {"touchdown_rate": 0.87, "goal_hold_rate": 0.94}} for s in range(n_seeds)]
    cur = {"wall_to_stage4_s": 1.55, "wall_all_levels_s": 2.33, "seeds_reaching_stage4_by_rule": n_seeds - 1, "n_seeds": n_seeds, "wall_to_stage4_by_rule_s": 1.41, "mode": "m" * 150, "workload": "c" * 100, "envs_per_gpu": 32768, "global_envs": 32768 * n_gpus,
           "episode_budget_per_level": 12582912, "sync_period": 16, "trainer_kw": {"quirks": 96}, "seeds": list(range(n_seeds)),
           "promoted_levels_per_seed": [5] * n_seeds, "level0_promoted_per_seed": [True] * n_seeds,
           "population_success_at_promotion": {"min": 0.95, "mean": 0.96, "note": "n" * 200}, "rule": "r" * 150,
           "stage4_greedy_4096_episodes": {"trained_mean": {"touchdown_rate": 0.87, "goal_hold_rate": 0.94}, "trained_worst_seed": {"touchdown_rate": 0.83, "goal_hold_rate": 0.9},
                                           "reference_assets": {"touchdown_rate": 0.8757, "goal_hold_rate": 0.954}},
           "attempts": {"max": 6, "accept_touchdown": 0.875, "per_seed": [1 + s % 3 for s in range(n_seeds)], "accepted_per_seed": [True] * n_seeds, "selection": "t" * 300,
                        "first_attempt": {"promoted_levels_per_seed": [5] * (n_seeds - 1) + [4], "trained_mean": {"touchdown_rate": 0.877, "goal_hold_rate": 0.949},
                                          "trained_worst_seed": {"touchdown_rate": 0.824, "goal_hold_rate": 0.935}}}, "runs": runs}
    full = {"metric": "env-steps/sec (whole node)", "value": 5.2e9 * n_gpus, "unit": "env-steps/s", "n_gpus": n_gpus, "steps": 2000, "warmup": 200, "preroll_steps": 512,
            "ms_per_step": 0.0247, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[4] share: " + "s" * 300, "workload_long": "l" * 600, "baseline_config": 4, "envs_per_gpu": 131072, "global_envs": 131072 * n_gpus,
                       "sync_period": 16 if n_gpus > 1 else 1, "exchange_rehearsal": False, "periods_per_launch": 16, "fold_per_step": 1, "parallelism": f"env-shard x{n_gpus}",
                       "block": 0, "two_axis": 0, "randomize_platform": 1, "noise": 1, "algorithmic_bytes_per_env_step": 328, "library_source_sha16": "0123456789abcdef"},
            "env_steps": 257611704, "device_ms_per_step": 0.0247, "roofline": roof, "valu_issue": issue,
            "repeats": {"n": 7, "value_min": 5.1e9 * n_gpus, "value_max": 5.3e9 * n_gpus, "ms_per_step_min": 0.0243, "ms_per_step_max": 0.0251, "statistic": "median over back-to-back repetitions of the K-step timed region"},
            "reference_quoted": {"reference+gazebo_env_steps_per_s": 20.18, "realtime_ceiling": 22.92, "reference_python_mdp+agent_us_per_step": 69.6,
                                 "reference_python_mdp+agent_steps_per_s": 14400.0, "source": "q" * 200},
            "curriculum": cur, "promoted_levels": [5] * n_seeds, "goal_hold_rate": 0.94, "touchdown_rate": 0.87, "wall_to_stage4_s": 1.55}
    if n_gpus == 1:
        full["small_batch"] = block(4096)
        full["large_batch"] = block(1048576)
        full["f64"] = dict(block(131072), dtype="f64")
        full["eps_0p1"] = {"eps": 0.1, "value": 5.6e9, "ms_per_step": 0.0231, "kernel_avg_ms": 0.369, "value_min": 5.5e9, "value_max": 5.7e9, "env_steps": 2.6e8, "roofline_frac": 0.23, "note": "n" * 200}
        full["cpu_baseline"] = {"value": 8.5e6, "unit": "env-steps/s", "cores": 16, "kind": "port", "sample": "p" * 400, "single_thread_value": 8.1e5, "single_thread_sample": "t" * 100}
    else:
        full["sync"] = {"exchange_name": "rccl", "sync_period": 16, "ms_per_step": 0.025, "ms_per_step_no_exchange": 0.0247, "sync_ms_per_step": 0.0004, "exchange_device_ms": 0.02, "exchanges_timed": 4,
                        "staleness_bound_periods": 48, "staleness_note": "a" * 300, "exchange": "e" * 200, "p2p_failed": False, "replicas_identical": True,
                        "p2p": {"value": 4.3e10, "ms_per_step": 0.0243, "sync_ms_per_step": 0.0002, "exchange_device_ms": 0.016, "exchanges_timed": 4, "replicas_identical": True,
                                "value_min": 4.2e10, "value_max": 4.4e10, "exchange": "e" * 200},
                        **{f"sync_period_{k}": {"value": 1e9, "ms_per_step": 0.03, "sync_ms_per_step": 0.01} for k in (2, 32)}}
        full["sync_ms_per_step"] = 0.0004
    return full


def test_line_is_compact_and_round_trips():
    for n_seeds, n_gpus in ((12, 1), (12, 8), (2, 8), (64, 1)):
        full = synthetic_record(n_seeds, n_gpus)
        assert len(json.dumps(full)) > 8192 or n_seeds == 2  # what round 3 printed
        line = json.dumps(bench.compact_line(full))
        assert len(line) < bench.LINE_LIMIT == 7000 < 8192, len(line)   # (the driver keeps the last 8 KB of stdout)
        rec = json.loads(line)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
            assert k in rec, k
        assert rec["value"] == full["value"] and rec["n_gpus"] == n_gpus
        assert rec["config"]["workload"].startswith("configs[4] share") and len(rec["config"]["workload"]) <= 160
        assert "model" not in rec["config"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "hbm_real_frac", "valu_issue_frac", "frac_is"):
            assert k in rec["roofline"], k
        assert rec["roofline"]["frac"] == full["roofline"]["frac"] and rec["roofline"]["bound"] == "valu_issue"   # the roof that binds
        assert rec["repeats"]["n"] == 7 and rec["repeats"]["value_min"] <= rec["value"] <= rec["repeats"]["value_max"]
        assert rec["curriculum"]["seeds_reaching_stage4_by_rule"] == n_seeds - 1 and rec["curriculum"]["wall_to_stage4_by_rule_s"] == 1.41
        assert rec["curriculum"]["promoted_levels_per_seed"] == [5] * n_seeds and "runs" not in rec["curriculum"]
        assert rec["curriculum"]["population_success_at_promotion"] == {"min": 0.95, "mean": 0.96}
        att = rec["curriculum"]["attempts"]   # several whole curricula per seed: what was needed, and the first attempts beside the chosen ones
        assert att["max"] == 6 and att["accept_touchdown"] == 0.875 and len(att["per_seed"]) == n_seeds and "selection" not in att
        assert att["first_attempt"]["trained_worst_seed"]["touchdown_rate"] == 0.824 and att["first_attempt"]["promoted_levels_per_seed"][-1] == 4
        assert rec["detail_file"] == bench.DETAIL_FILE
        if n_gpus == 1:
            assert set(("value", "unit", "cores", "kind", "sample")) <= set(rec["cpu_baseline"])
            assert rec["small_batch"]["roofline_frac"] == 0.214 and rec["large_batch"]["envs"] == 1048576
            assert rec["valu_issue"]["frac_at_measured_clock"] <= 1.0
            assert rec["f64"]["envs"] == 131072 and "kernel_avg_ms" in rec["f64"] and rec["f64"]["roofline"]["bound"] == "valu_issue"
            assert rec["eps_0p1"]["eps"] == 0.1 and "roofline_frac" in rec["eps_0p1"]
        else:
            assert rec["sync"]["staleness_bound_periods"] == 48 and "sync_period_2" not in rec["sync"]
            assert rec["sync"]["replicas_identical"] is True and rec["sync"]["exchange_name"] == "rccl"
            assert rec["sync"]["p2p"]["replicas_identical"] is True and "sync_ms_per_step" in rec["sync"]["p2p"]   # both exchanges in ONE run


def test_line_of_the_committed_round3_record_fits():
    full = json.loads((ROOT / "profiles" / "r3_bench_default.json").read_text())
    assert len(json.dumps(full)) > 20000
    line = json.dumps(bench.compact_line(full))
    assert len(line) < 7000
    rec = json.loads(line)
    assert rec["roofline"]["kernel_avg_ms"] == full["roofline"]["kernel_avg_ms"] and rec["cpu_baseline"]["cores"] == full["cpu_baseline"]["cores"]


def test_emit_writes_the_detail_file(tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", tmp_path)
    full = synthetic_record(12, 1)
    line = bench.emit(full)
    assert "\n" not in line and len(line) < 7000
    assert json.loads((tmp_path / bench.DETAIL_FILE).read_text()) == json.loads(json.dumps(full))


def test_gpus2_relay_forwards_rank0_line_unchanged(capfd):
    """`bench.py --gpus 2` without a launcher starts its ranks itself and relays rank 0's line: stub children stand in for the ranks"""
    line = json.dumps(bench.compact_line(synthetic_record(2, 2)))
    child = ("import os, sys\n"
             "if os.environ['RANK'] == '0':\n"
             "    print('NCCL banner line on stdout')\n"
             f"    print({line!r})\n"
             "else:\n"
             "    print('rank 1 chatter', file=sys.stderr)\n")
    rc = bench.spawn_ranks(SimpleNamespace(gpus=2), child_argv=[sys.executable, "-c", child])
    out, err = capfd.readouterr()
    assert rc == 0
    assert out.strip().splitlines()[-1] == line          # unchanged, last thing on stdout
    assert "rank 1 chatter" not in err                    # ranks != 0 are quiet unless the job fails
    # a rank that dies: non-zero exit, its stderr shown, no line relayed
    bad = ("import os, sys\n"
           "if os.environ['RANK'] == '1':\n"
           "    print('no GPU for rank 1', file=sys.stderr); sys.exit(3)\n"
           "import time; time.sleep(30)\n")
    rc = bench.spawn_ranks(SimpleNamespace(gpus=2), child_argv=[sys.executable, "-c", bad])
    out, err = capfd.readouterr()
    assert rc == 1 and out.strip() == "" and "no GPU for rank 1" in err


def test_default_seed_count_depends_on_gpus():
    src = (ROOT / "bench.py").read_text()
    assert "args.curriculum_seeds = 12 if args.gpus == 1 else 2" in src
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "--curriculum-seeds" in r.stdout


def test_measured_clock_ignores_oversubscribed_batches():
    """at 1 M envs 16 waves share a SIMD and do not all run at once: the phase-clock file's cycles / time is 0.5 "GHz" there — not a clock, and a
    fraction priced with it would exceed 1 (round 4's first campaign printed 3.4)"""
    ghz, _ = bench.committed_clock(1048576, True)
    assert ghz is None
    ghz, src = bench.committed_clock(131072, True)
    assert ghz is not None and 1.5 < ghz < 2.5 and src.startswith("r")
    v = bench.valu_issue(1048576, 16, 0, 1, 1, "f32", 2.04, 16)
    assert v is None or v.get("frac_at_measured_clock") is None or v["frac_at_measured_clock"] <= 1.0
