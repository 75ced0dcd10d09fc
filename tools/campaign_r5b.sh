set -e
# round 5, second GPU call: parity suite on the new default library, then in-run A/B of the constants' placement on the headline workload
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
REPS=3 bash tools/ab_bench.sh "base nolitm default" --no-f64-block > $O/ab_headline.txt 2>&1 || { tail $O/ab_headline.txt; exit 1; }
cat $O/ab_headline.txt
REPS=2 bash tools/ab_bench.sh "base default" --config 1 --no-f64-block > $O/ab_config1.txt 2>&1; cat $O/ab_config1.txt
REPS=2 bash tools/ab_bench.sh "base default" --config 3 --no-f64-block > $O/ab_config3.txt 2>&1; cat $O/ab_config3.txt
REPS=2 bash tools/ab_bench.sh "base default" --envs 1048576 --steps 320 --no-f64-block > $O/ab_1m.txt 2>&1; cat $O/ab_1m.txt
python bench.py --gpus 1 --steps 20 --warmup 5 --no-curriculum > $O/bench_driver_args.json 2> $O/bench.err || { tail $O/bench.err; exit 1; }
cp bench_detail.json $O/bench_driver_args_detail.json
python bench.py --exchange-rehearsal --no-curriculum --no-cpu-baseline > $O/bench_exchange_rehearsal.json 2>> $O/bench.err || { tail $O/bench.err; exit 1; }
cp bench_detail.json $O/bench_exchange_rehearsal_detail.json
cat $O/bench_driver_args.json; cat $O/bench_exchange_rehearsal.json
