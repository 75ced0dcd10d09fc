set -e
# usage: bash tools/campaign_r5.sh [bench|pmc|all]   (on the GPU box, from the repo root; results under gpurun_out/r5p; tools/collect_r5.sh copies them into profiles/)
WHAT=${1:-all}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5p
mkdir -p $O
cd $R
if [ $WHAT != pmc ]; then
python bench.py > $O/bench_default.json 2> $O/bench_default.err
cp bench_detail.json $O/bench_default_detail.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2>> $O/bench_default.err
cp bench_detail.json $O/bench_driver_args_detail.json
echo bench done
for c in 1 2 3; do python bench.py --config $c --no-curriculum --no-cpu-baseline --small-envs 0 --large-envs 0 > $O/bench_config$c.json 2>> $O/bench.err; done
HSA_ENABLE_IPC_MODE_LEGACY=0 python bench.py --exchange-rehearsal --no-curriculum --no-cpu-baseline > $O/bench_exchange_rehearsal.json 2>> $O/bench.err
cp bench_detail.json $O/bench_exchange_rehearsal_detail.json
python scripts/plumbing_config1.py > $O/config1_plumbing.json 2>> $O/bench.err
echo configs done
if [ -f dql_multirotor_landing_amd/csrc/libdql_hip_phase.so ]; then
  DQL_LIB_PATH=$R/dql_multirotor_landing_amd/csrc/libdql_hip_phase.so python tools/exp_phase_clock.py 4096,32768,131072,1048576 cfg4 > $O/phase_clock.jsonl 2>> $O/bench.err
  DQL_LIB_PATH=$R/dql_multirotor_landing_amd/csrc/libdql_hip_phase.so python tools/exp_phase_clock.py 4096,131072 >> $O/phase_clock.jsonl 2>> $O/bench.err
fi
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --small-envs 0 --large-envs 0 --no-curriculum > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c1 -- python3 $R/bench.py --config 1 --no-cpu-baseline --small-envs 0 --large-envs 0 --no-curriculum > $O/bench_c1_under_rocprof.json 2>> $O/rocprof_stats.err
echo stats done
fi
if [ $WHAT != bench ]; then
rm -rf $O/pmc_*
for cfg in "4096 640 _" "32768 480 _" "131072 320 cfg4" "1048576 160 cfg4"; do
  set -- $cfg
  F=""; S=""; if [ $3 != _ ]; then F=$3; S=_$3; fi
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES -d $O/pmc_sq_$1_p16$S -- python3 $R/tools/prof_run.py $1 $2 0 16 $F > /dev/null 2>> $O/pmc.err
  rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch_$1_p16$S -- python3 $R/tools/prof_run.py $1 $2 0 16 $F > /dev/null 2>> $O/pmc.err
  rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmc_write_$1_p16$S -- python3 $R/tools/prof_run.py $1 $2 0 16 $F > /dev/null 2>> $O/pmc.err
  echo pmc $1 done
done
# where a wave's cycles go at the headline batch: issue, waits, transcendental share — two passes (counter groups)
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM -d $O/pmcx_a_131072_p16_cfg4 -- python3 $R/tools/prof_run.py 131072 320 0 16 cfg4 > /dev/null 2>> $O/pmc.err || true
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_TRANS SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAVES -d $O/pmcx_b_131072_p16_cfg4 -- python3 $R/tools/prof_run.py 131072 320 0 16 cfg4 > /dev/null 2>> $O/pmc.err || true
echo pmcx done
fi
cd $R
# keep only the csv summaries (the merge back is capped at 64 MiB)
find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete
du -sh $O
