"""The committed measurement files describe the committed sources and agree with each other (CPU, no GPU): the bench line's kernel time against the
rocprofv3 `--kernel-trace --stats` summary of the same command, the counter files' source stamp against the library sources in the tree, the roofline
arithmetic of the line against its own fields."""
import csv
import json
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
P = ROOT / "profiles"
# the latest round that committed a counter pass
ROUND = max((f.name.split("_")[0] for f in P.glob("r*_traffic.json")), key=lambda r: int(r[1:]))
HOW = ("the kernel sources changed after the committed measurement: rebuild the phase library (tools/ab_build.sh phase -DDQL_PHASE_CLOCK), run `bash tools/campaign_r5.sh pmc` on the GPU box, "
       "`DQL_ROUND=r5 python tools/pmc_summary.py gpurun_out/r5p/pmc_*`, then `bash tools/campaign_r5.sh bench` and `bash tools/collect_r5.sh` (tools/README.md)")


def _line(name):
    return json.loads((P / f"{ROUND}_{name}.json").read_text())


def test_counter_files_and_bench_lines_are_stamped_with_the_tree_s_kernel_sources():
    import bench
    sha = bench.lib_source_sha16()
    assert json.loads((P / f"{ROUND}_traffic.json").read_text())["source_sha16"] == sha, HOW
    assert json.loads((P / f"{ROUND}_pmc_sq_summary.json").read_text())["source_sha16"] == sha, HOW
    for name in ("bench_default", "bench_driver_args"):
        d = _line(name)
        assert d["config"]["library_source_sha16"] == sha, f"{name}: {HOW}"
        assert d["roofline"]["traffic"] is not None, name     # (null = the counter pass belongs to other sources: bench.py drops it then)


def test_kernel_time_of_the_line_agrees_with_rocprofv3_stats_of_the_same_command():
    d = _line("bench_default")
    rows = list(csv.DictReader(open(P / f"{ROUND}_bench_kernel_stats_config4.csv")))
    k = [r for r in rows if r["Name"].startswith("void k_step<float, 256, 3, 1>")]
    assert len(k) == 1 and int(k[0]["Calls"]) > 100
    avg_ms = float(k[0]["AverageNs"]) * 1e-6
    assert abs(avg_ms - d["roofline"]["kernel_avg_ms"]) / avg_ms < 0.02          # HIP events in bench.py vs the profiler's kernel trace: within 2 %
    assert float(k[0]["Percentage"]) > 90.0                                       # the dominant kernel IS the one the roofline block is about


def test_roofline_block_is_its_own_arithmetic():
    for name in ("bench_default", "bench_driver_args"):
        d = _line(name); r = d["roofline"]
        assert r["bound"] == "valu_issue" and r["frac_is"] == "hbm_algorithmic" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
        achieved = r["algorithmic_bytes_per_env_step"] * r["env_steps_per_launch"] / (r["kernel_avg_ms"] * 1e-3) / 1e9
        assert abs(achieved - r["achieved"]) / achieved < 1e-6 and abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-9
        assert abs(r["traffic"] / (r["kernel_avg_ms"] * 1e-3) / 8e12 - r["hbm_real_frac"]) < 1e-6
        assert d["value"] == sorted([d["repeats"]["value_min"], d["value"], d["repeats"]["value_max"]])[1] and d["repeats"]["n"] >= 7
        assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1 and d["vs_baseline"] is None
