set -e
# usage: bash tools/campaign_r2.sh [bench|pmc|all]   (on the GPU box, from the repo root; results under gpurun_out/r2p)
WHAT=${1:-all}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2p
mkdir -p $O
cd $R
if [ $WHAT != pmc ]; then
python bench.py > $O/bench_4096.json 2> $O/bench_4096.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_4096_driver_args.json 2>> $O/bench_4096.err
echo bench done
python bench.py --envs 32768 --no-curriculum --no-cpu-baseline --large-envs 0 > $O/bench_32768.json 2>> $O/bench.err
python bench.py --envs 131072 --randomize-platform 1 --noise 1 --steps 1000 --warmup 100 --no-curriculum --no-cpu-baseline --large-envs 0 > $O/bench_config5_131072.json 2>> $O/bench.err
python bench.py --envs 1048576 --steps 300 --warmup 32 --no-curriculum --no-cpu-baseline --large-envs 0 > $O/bench_1M.json 2>> $O/bench.err
python bench.py --envs 65536 --two-axis 1 --steps 1000 --warmup 100 --no-cpu-baseline --large-envs 0 > $O/bench_2axis_65536.json 2>> $O/bench.err
echo sizes done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --large-envs 0 --no-curriculum > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err
echo stats done
fi
if [ $WHAT != bench ]; then
rm -rf $O/pmc_*
for cfg in "4096 400" "32768 400" "131072 240" "1048576 120"; do
  set -- $cfg
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES -d $O/pmc_sq_$1_p8 -- python3 $R/tools/prof_run.py $1 $2 0 8 > /dev/null 2>> $O/pmc.err
  rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch_$1_p8 -- python3 $R/tools/prof_run.py $1 $2 0 8 > /dev/null 2>> $O/pmc.err
  rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmc_write_$1_p8 -- python3 $R/tools/prof_run.py $1 $2 0 8 > /dev/null 2>> $O/pmc.err
  echo pmc $1 done
done
fi
cd $R
# keep only the csv summaries (the merge back is capped at 64 MiB)
find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete
tools/micro/valu_forms > $O/valu_forms.jsonl
du -sh $O
