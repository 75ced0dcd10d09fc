set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5d; mkdir -p $O
B="quirks=96,judge_envs=64,ppl=16,eps_tail=0.0,eps_tail_after=192,population_gate=0.94,sync_period=16"
python tools/exp_curriculum.py --envs 32768 --budget-per-env 768 --seeds 42 1 2 3 4 5 6 7 8 9 10 11 --set "base:$B" "r96:$B,restart_after=96" "r64:$B,restart_after=64" > $O/curr_12.jsonl 2> $O/curr.err || { tail $O/curr.err; exit 1; }
python - <<'PY'
import json, collections
agg=collections.defaultdict(list)
for l in open('gpurun_out/r5d/curr_12.jsonl'):
    d=json.loads(l); agg[d['set']].append(d)
    print(d['set'], d['seed'], d['promoted_levels'], d['goal_hold'], d['touchdown'], d['wall_s'], [(x['level'], x['promoted'], x['pop'], x['episodes_per_env']) for x in d['levels']])
for k,v in agg.items():
    print(k, 'all5', sum(1 for d in v if d['promoted_levels']==5), 'levels', [d['promoted_levels'] for d in v], 'goal mean %.3f min %.3f' % (sum(d['goal_hold'] for d in v)/len(v), min(d['goal_hold'] for d in v)), 'td mean %.3f min %.3f' % (sum(d['touchdown'] for d in v)/len(v), min(d['touchdown'] for d in v)))
PY
