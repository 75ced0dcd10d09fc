set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5j; mkdir -p $O
B="judge_envs=64,ppl=16,eps_tail=0.0,eps_tail_after=192,population_gate=0.94,sync_period=16,restart_after=96,quirks=96"
python tools/exp_curriculum.py --envs 32768 --budget-per-env 768 --seeds 8 10 3 5 6 --set "sb1m6:$B,step_back_after=1,max_step_backs=6" "sb1m10:$B,step_back_after=1,max_step_backs=10" > $O/curr_step_back_more.jsonl 2> $O/curr.err || { tail $O/curr.err; exit 1; }
python - <<'PY'
import json, collections
agg=collections.defaultdict(list)
for l in open('gpurun_out/r5j/curr_step_back_more.jsonl'):
    d=json.loads(l); agg[d['set']].append(d)
for k,v in agg.items():
    print(k, 'all5', sum(1 for d in v if d['promoted_levels']==5), 'levels', [d['promoted_levels'] for d in v], 'goal mean %.3f min %.3f' % (sum(d['goal_hold'] for d in v)/len(v), min(d['goal_hold'] for d in v)), 'td mean %.3f min %.3f' % (sum(d['touchdown'] for d in v)/len(v), min(d['touchdown'] for d in v)), 'stage4 wall', round(sum(d['wall_to_stage4_s'] or 0 for d in v)/len(v),2), 'wall', round(sum(d['wall_s'] for d in v)/len(v),2))
    for d in v:
        if d['promoted_levels']<5 or any(x['step_backs'] for x in d['levels']): print('   ', d['seed'], d['wall_s'], d['touchdown'], d['goal_hold'], [(x['level'], x['promoted'], x['pop'], x['restarts'], x['step_backs']) for x in d['levels']])
PY
