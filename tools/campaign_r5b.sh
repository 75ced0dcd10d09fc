set -e
# round 5, second GPU call: parity suite on the new default library, then in-run A/B of the constants' placement on the headline workload
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
REPS=3 bash tools/ab_bench.sh "base nolitm default" > $O/ab_headline.txt 2>&1 || { tail $O/ab_headline.txt; exit 1; }
cat $O/ab_headline.txt
REPS=2 bash tools/ab_bench.sh "base nolitm default" --config 1 > $O/ab_config1.txt 2>&1; cat $O/ab_config1.txt
REPS=2 bash tools/ab_bench.sh "base nolitm default" --config 3 > $O/ab_config3.txt 2>&1; cat $O/ab_config3.txt
REPS=2 bash tools/ab_bench.sh "base default" --envs 1048576 --steps 320 > $O/ab_1m.txt 2>&1; cat $O/ab_1m.txt
