set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5h; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "periods_per_launch" > $O/pytest_p32.log 2>&1 || { tail -30 $O/pytest_p32.log; exit 1; }
tail -2 $O/pytest_p32.log
for P in 16 32; do for K in 2000 20; do
python bench.py --periods-per-launch $P --steps $K --warmup $([ $K = 20 ] && echo 5 || echo 200) --no-f64-block --no-cpu-baseline --small-envs 0 --large-envs 0 --no-curriculum | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('P $P K $K', '%.4g env-steps/s' % d['value'], '%.2f us/period' % (d['ms_per_step'] * 1e3), 'frac %.4f' % d['roofline']['frac'], d['repeats']['value_min'], d['repeats']['value_max'])
"; done; done > $O/bench_p32.txt 2>&1
cat $O/bench_p32.txt
B="quirks=96,judge_envs=64,eps_tail=0.0,eps_tail_after=192,population_gate=0.94,restart_after=96"
python tools/exp_curriculum.py --envs 32768 --budget-per-env 768 --seeds 42 1 2 3 4 5 6 7 8 9 10 11 --set "p32s32:$B,ppl=32,sync_period=32" > $O/curr_p32.jsonl 2> $O/curr.err || { tail $O/curr.err; exit 1; }
python - <<'PY'
import json, collections
agg=collections.defaultdict(list)
for l in open('gpurun_out/r5h/curr_p32.jsonl'):
    d=json.loads(l); agg[d['set']].append(d)
for k,v in agg.items():
    print(k, 'all5', sum(1 for d in v if d['promoted_levels']==5), 'levels', [d['promoted_levels'] for d in v], 'goal mean %.3f min %.3f' % (sum(d['goal_hold'] for d in v)/len(v), min(d['goal_hold'] for d in v)), 'td mean %.3f min %.3f' % (sum(d['touchdown'] for d in v)/len(v), min(d['touchdown'] for d in v)), 'stage4 wall', round(sum(d['wall_to_stage4_s'] or 0 for d in v)/len(v),2))
PY
