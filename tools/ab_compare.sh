for v in "" _ilp _iter; do
  if [ -f dql_multirotor_landing_amd/csrc/libdql_hip$v.so ]; then
  DQL_LIB_PATH=$PWD/dql_multirotor_landing_amd/csrc/libdql_hip$v.so python tools/sweep_ppl.py 4096,32768,131072,1048576 8 0 0 | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$v', d['envs'], '%.2f us' % d['us_per_period'])
"; fi; done
