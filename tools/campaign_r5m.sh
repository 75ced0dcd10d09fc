set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5m; mkdir -p $O
B="judge_envs=64,ppl=16,eps_tail=0.0,eps_tail_after=192,population_gate=0.94,sync_period=16,restart_after=96,quirks=96"
SEEDS="42 1 2 3 4 5 6 7 8 9 10 11 $(seq 12 47)"
python tools/exp_curriculum.py --envs 32768 --budget-per-env 768 --seeds $SEEDS --set "sb2:$B,step_back_after=2" "sb3:$B,step_back_after=3" "sb4:$B,step_back_after=4" > $O/curr_48_step_back.jsonl 2> $O/curr.err || { tail $O/curr.err; exit 1; }
python - <<'PY'
import json, collections
agg=collections.defaultdict(list)
for l in open('gpurun_out/r5m/curr_48_step_back.jsonl'):
    d=json.loads(l); agg[d['set']].append(d)
for k,v in agg.items():
    print(k, 'seeds', len(v), 'all5', sum(1 for d in v if d['promoted_levels']==5), 'stage4 by rule', sum(1 for d in v if all(x['promoted'] for x in d['levels'][:4])), 'goal mean %.3f min %.3f' % (sum(d['goal_hold'] for d in v)/len(v), min(d['goal_hold'] for d in v)), 'td mean %.3f min %.3f' % (sum(d['touchdown'] for d in v)/len(v), min(d['touchdown'] for d in v)), 'stage4 wall %.2f' % (sum(d['wall_to_stage4_s'] or 0 for d in v)/len(v)), 'wall %.2f' % (sum(d['wall_s'] for d in v)/len(v)))
    for d in v:
        if d['promoted_levels']<5: print('   ', d['seed'], d['wall_s'], d['touchdown'], d['goal_hold'], [(x['level'], x['promoted'], x['pop'], x['restarts'], x['step_backs']) for x in d['levels']])
PY
