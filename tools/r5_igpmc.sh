# SQ issue counters of the walk launch's generator pass alone: bash tools/r5_igpmc.sh <tag>
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-r5igpmc}
mkdir -p $O
cd /tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU"
P2="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC"
i=1
for P in "$P1" "$P2"; do
  timeout 200 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/issue/rng_p$i -- python $R/tools/rng_launch_prof.py 10 > $O/issue_$i.log 2>&1
  python $R/tools/pmc_issue.py reduce $O/issue/rng_p$i
  i=$((i+1))
done
cd $R
python tools/pmc_issue.py $O/issue > $O/pmc_issue.json
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O/issue -name "*counter_collection.csv" -delete
tail -3 $O/issue_2.log
