#!/bin/bash
# training-dynamics experiments (paper mode): fold semantics x eps floor x envs
cd "$(dirname "$0")/.."
for cfg in "4096 0 0.0" "4096 1 0.0" "4096 1 0.02" "4096 0 0.02" "16384 1 0.02" "1024 1 0.02"; do
  set -- $cfg
  echo "=== envs=$1 fold_per_step=$2 eps_floor=$3"
  timeout -k 10 200 python scripts/training.py --envs $1 --mode paper --out gpurun_out/exp_train_$1_$2_$3 --max-steps-per-level 30000 --max-episodes 1000000000 --chunk 256 --fold-per-step $2 --eps-floor $3 2>&1 | python -c "
import sys, json
d=json.loads(sys.stdin.read())
for h in d['history']: print({k:(round(v,3) if isinstance(v,float) else v) for k,v in h.items()})"
done
