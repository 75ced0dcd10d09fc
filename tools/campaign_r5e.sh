set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5e; mkdir -p $O
B="quirks=96,ppl=16,sync_period=16,as_launched=1"
python tools/exp_curriculum.py --envs 32768 --budget-per-env 768 --seeds 42 1 2 3 4 5 6 7 8 9 10 11 --set \
  "al_j64:$B,judge_envs=64,eps_tail=0.0,eps_tail_after=192" \
  "al_j64_scale4096:$B,judge_envs=64,eps_episode_scale=4096,eps_tail=0.0,eps_tail_after=400" \
  > $O/curr_al12.jsonl 2> $O/curr.err || { tail $O/curr.err; exit 1; }
python - <<'PY'
import json, collections
agg=collections.defaultdict(list)
for l in open('gpurun_out/r5e/curr_al12.jsonl'):
    d=json.loads(l); agg[d['set']].append(d)
for k,v in agg.items():
    print(k, 'level0 promoted', sum(1 for d in v if d['levels'][0]['promoted']), 'of', len(v), 'pop at level0', [d['levels'][0]['pop'] for d in v], 'levels', [d['promoted_levels'] for d in v], 'goal', [d['goal_hold'] for d in v])
PY
