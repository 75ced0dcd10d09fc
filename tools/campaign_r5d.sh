set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5d; mkdir -p $O
B="quirks=96,judge_envs=64,ppl=16,eps_tail=0.0,eps_tail_after=192,population_gate=0.94,sync_period=16,restart_after=96"
python tools/exp_curriculum.py --envs 32768 --budget-per-env 768 --seeds 8 10 42 5 3 --set "deep2:$B,deep_restart_every=2" "deep1:$B,deep_restart_every=1" > $O/curr_deep.jsonl 2> $O/curr.err || { tail $O/curr.err; exit 1; }
python - <<'PY'
import json
for l in open('gpurun_out/r5d/curr_deep.jsonl'):
    d=json.loads(l); print(d['set'], d['seed'], d['promoted_levels'], d['goal_hold'], d['touchdown'], d['wall_s'], [(x['level'], x['promoted'], x['pop'], x['episodes_per_env']) for x in d['levels']])
PY
