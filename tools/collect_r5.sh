# copies the campaign's results (gpurun_out/r5p) into profiles/ : bash tools/collect_r5.sh
set -e
cd "$(dirname "$0")/.."
DQL_ROUND=r5 python tools/pmc_summary.py gpurun_out/r5p/pmc_* | tail -1
cp "$(ls -t gpurun_out/r5p/stats/*/*kernel_stats.csv gpurun_out/r5p/stats/*kernel_stats.csv 2>/dev/null | head -1)" profiles/r5_bench_kernel_stats_config4.csv
cp "$(ls -t gpurun_out/r5p/stats_c1/*/*kernel_stats.csv gpurun_out/r5p/stats_c1/*kernel_stats.csv 2>/dev/null | head -1)" profiles/r5_bench_kernel_stats_config1.csv
for f in bench_default bench_default_detail bench_driver_args bench_driver_args_detail bench_config1 bench_config2 bench_config3 bench_exchange_rehearsal bench_exchange_rehearsal_detail config1_plumbing; do cp gpurun_out/r5p/$f.json profiles/r5_$f.json; done
[ -f gpurun_out/r5p/phase_clock.jsonl ] && cp gpurun_out/r5p/phase_clock.jsonl profiles/r5_phase_clock.jsonl
python tools/pmcx_summary.py gpurun_out/r5p/pmcx_a_131072_p16_cfg4 gpurun_out/r5p/pmcx_b_131072_p16_cfg4 > profiles/r5_pmc_wave_cycles_131072_cfg4.json || true
