set -e
# end of round 5: the PMC passes of tools/campaign_r5a.sh on the FINAL kernel; results under gpurun_out/r5o, summarised into
# profiles/r5_pmc_wave_cycle_breakdown.json ("end_of_round5") by tools/pmc_breakdown.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5o
rm -rf $O; mkdir -p $O
W="python3 $R/tools/prof_run.py 131072 320 0 16 cfg4"
p() { name=$1; shift; rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $O/pmc_$name -- $W > /dev/null 2>> $O/pmc.err || echo "pass $name failed" >> $O/pmc.err; echo pass $name; }
p a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES
p b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_WAVES
p c SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAVES
p g SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT
p h SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_TRANS_F64 SQ_WAVES
cd $R
find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete
du -sh $O
