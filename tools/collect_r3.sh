# copies the campaign's results (gpurun_out/r3p) into profiles/ : bash tools/collect_r3.sh
set -e
cd "$(dirname "$0")/.."
DQL_ROUND=r3 python tools/pmc_summary.py gpurun_out/r3p/pmc_* | tail -1
cp "$(ls -t gpurun_out/r3p/stats/runc/*kernel_stats.csv | head -1)" profiles/r3_bench_kernel_stats_config4.csv
cp "$(ls -t gpurun_out/r3p/stats_c1/runc/*kernel_stats.csv | head -1)" profiles/r3_bench_kernel_stats_config1.csv
for f in bench_default bench_driver_args bench_config1 bench_config2 bench_config3 bench_exchange_rehearsal bench_exchange_rehearsal_p2p; do cp gpurun_out/r3p/$f.json profiles/r3_$f.json; done
