#!/bin/bash
# in-run A/B of library variants built by tools/ab_build.sh on the headline workload: tools/ab_bench.sh "name1 name2 ..." [bench args]
# ("" = the default library); every variant is timed twice, interleaved, inside ONE gpurun call (box-to-box differences are 2-3 %)
V="$1"; shift
for rep in $(seq 1 ${REPS:-2}); do
  for v in $V; do
    L=dql_multirotor_landing_amd/csrc/libdql_hip_$v.so; [ "$v" = default ] && L=dql_multirotor_landing_amd/csrc/libdql_hip.so
    DQL_LIB_PATH=$PWD/$L python bench.py --no-cpu-baseline --small-envs 0 --large-envs 0 --no-curriculum "$@" | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('$v', 'rep$rep', '%.4g env-steps/s' % d['value'], '%.2f us/period' % (d['ms_per_step'] * 1e3), d['roofline'].get('kernel'))
"
  done
done
