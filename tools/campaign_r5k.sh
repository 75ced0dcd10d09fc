set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5k; mkdir -p $O
B="judge_envs=64,ppl=16,eps_tail=0.0,eps_tail_after=192,population_gate=0.94,sync_period=16,restart_after=96,step_back_after=1,quirks=96"
python tools/exp_curriculum.py --envs 32768 --budget-per-env 768 --seeds 42 1 2 3 4 5 6 7 8 9 10 11 --set "fc3:$B,final_candidates=3" "fc5:$B,final_candidates=5" > $O/curr_final_candidates.jsonl 2> $O/curr.err || { tail $O/curr.err; exit 1; }
python - <<'PY'
import json, collections
agg=collections.defaultdict(list)
for l in open('gpurun_out/r5k/curr_final_candidates.jsonl'):
    d=json.loads(l); agg[d['set']].append(d)
for k,v in agg.items():
    print(k, 'all5', sum(1 for d in v if d['promoted_levels']==5), 'levels', [d['promoted_levels'] for d in v], 'goal mean %.3f min %.3f' % (sum(d['goal_hold'] for d in v)/len(v), min(d['goal_hold'] for d in v)), 'td mean %.3f min %.3f' % (sum(d['touchdown'] for d in v)/len(v), min(d['touchdown'] for d in v)), 'stage4 wall', round(sum(d['wall_to_stage4_s'] or 0 for d in v)/len(v),2), 'wall', round(sum(d['wall_s'] for d in v)/len(v),2))
    for d in v:
        print('   ', d['seed'], d['wall_s'], 'td', d['touchdown'], 'gh', d['goal_hold'], d['levels'][-1].get('final_candidates'), d['levels'][-1].get('selected'), d['levels'][-1]['promoted'])
PY
