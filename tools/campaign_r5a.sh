set -e
# round 5, first GPU call: new operator tests + the whole GPU suite, the float32-vs-float64 per-field survey, and the PMC passes VERDICT r4 item 2 asks for
# (program directly after `--`; counters only with --kernel-trace).  Results under gpurun_out/r5a.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5a
mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
python tools/exp_f32_vs_f64.py 4096 > $O/f32_vs_f64.jsonl 2> $O/f32_vs_f64.err
echo survey done
cd /tmp
rocprofv3 -L > $O/counters.txt 2>&1 || true
W="python3 $R/tools/prof_run.py 131072 320 0 16 cfg4"
p() { name=$1; shift; rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $O/pmc_$name -- $W > /dev/null 2>> $O/pmc.err || echo "pass $name failed" >> $O/pmc.err; echo pass $name; }
p a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES
p b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_WAVES
p c SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAVES
p d SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_TRANS SQ_INSTS_SENDMSG SQ_WAVES
p e SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_IFETCH SQ_INSTS_FLAT SQ_INSTS_GDS SQ_WAVES
p f SQ_VALU_MFMA_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_WAVES
p g SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT
p h SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_TRANS_F64 SQ_WAVES
cd $R
find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete
du -sh $O
