#!/bin/bash
# round 5, attempts: bench.py's curriculum leg with up to 6 whole curricula per seed (first accepted kept), 12 seeds, and the training script's --attempts
set -e
mkdir -p gpurun_out/r5n
python3 bench.py --gpus 1 --steps 200 --warmup 20 --small-envs 0 --large-envs 0 --no-f64-block --cpu-steps 60 > gpurun_out/r5n/bench_attempts.json 2> gpurun_out/r5n/bench_attempts.err
cp bench_detail.json gpurun_out/r5n/bench_attempts_detail.json 2>/dev/null || true
python3 scripts/training.py --recipe bench --envs 32768 --attempts 4 --out gpurun_out/r5n/train_out > gpurun_out/r5n/training_attempts.json 2> gpurun_out/r5n/training_attempts.err
python3 -m pytest tests/test_gpu_dropin.py -q -m gpu -x -k "evaluation or simulation or greedy" > gpurun_out/r5n/pytest.log 2>&1
