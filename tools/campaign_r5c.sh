set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
REPS=3 bash tools/ab_bench.sh "base notan default" --no-f64-block > $O/ab_headline.txt 2>&1 || { tail $O/ab_headline.txt; exit 1; }
cat $O/ab_headline.txt
REPS=2 bash tools/ab_bench.sh "base default" --config 2 --no-f64-block > $O/ab_config2.txt 2>&1; cat $O/ab_config2.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c
W="python3 $R/tools/prof_run.py 131072 320 0 16 cfg4"
p() { name=$1; shift; rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $O/pmc_$name -- $W > /dev/null 2>> $O/pmc.err || echo "pass $name failed" >> $O/pmc.err; echo pass $name; }
p a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES
p b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_WAVES
p c SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAVES
find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete
