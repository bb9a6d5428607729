# Round-6 profile collection (run on the GPU box through gpurun): kernel stats of the bench's timed launch shape and of
# the device-resident loops (both RNG modes), HBM-traffic PMC passes for the bench AND for C4, SQ
# issue / stall PMC passes per workload (tools/pmc_issue.py) plus a per-SIMD busy pass (item 5: SQ_BUSY_CU_CYCLES, the
# vector / scalar instruction-cycle counters).  usage: bash tools/prof_r06.sh <outdir under gpurun_out> [quick]
set -x
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-prof_r06}
mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o bench -- python $R/bench.py --lean --steps 20 --warmup 5 > $O/bench_line_under_rocprof.json 2> $O/rocprof_bench.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o ns_c5 -- python $R/tools/r6_ns_modes.py pcg64 64 512 2 > $O/ns_c5.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o ns_c5_philox -- python $R/tools/r6_ns_modes.py philox 64 512 2 > $O/ns_c5_philox.log 2>&1
if [ -z "$2" ]; then
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o ns_c3 -- python $R/tools/ns_c3.py 16 > $O/ns_c3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o ns_c4 -- python $R/tools/c4_ksweep.py 16 128 > $O/ns_c4.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o ns_c1 -- python $R/tools/ns_c1.py 64 64 > $O/ns_c1.log 2>&1
fi
# HBM traffic: separate --pmc passes, --kernel-trace only (MI355X_MICROARCH.md)
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -- python $R/bench.py --lean --preroll 0 --steps 3 --warmup 1 > $O/pmc_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -- python $R/bench.py --lean --preroll 0 --steps 3 --warmup 1 > $O/pmc_w.log 2>&1
if [ -z "$2" ]; then
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_c4_f -- python $R/tools/c4_ksweep.py 16 128 > $O/pmc_c4_f.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_c4_w -- python $R/tools/c4_ksweep.py 16 128 > $O/pmc_c4_w.log 2>&1
fi
# issue / stall counters: SQ block, 8 a pass; pass 4 = the per-SIMD busy question of itemgen_kernel
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS"
P3="SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"
P4="SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INST_CYCLES_VALU SQ_IFETCH SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SALU"
i=1
for P in "$P1" "$P2" "$P3" "$P4"; do
  timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/issue/bench_p$i -- python $R/bench.py --lean --preroll 0 --steps 3 --warmup 1 > $O/issue_bench_$i.log 2>&1
  timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/issue/c5_p$i -- python $R/tools/ns_c5.py 64 512 once > $O/issue_c5_$i.log 2>&1
  if [ -z "$2" ]; then
  timeout 400 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/issue/c4_p$i -- python $R/tools/c4_ksweep.py 16 128 > $O/issue_c4_$i.log 2>&1
  fi
  for w in bench c5 c4; do [ -d $O/issue/${w}_p$i ] && python $R/tools/pmc_issue.py reduce $O/issue/${w}_p$i; done
  i=$((i+1))
done
cd $R
python tools/pmc_traffic.py $O/pmc_f $O/pmc_w > $O/pmc_traffic.json
[ -d $O/pmc_c4_f ] && python tools/pmc_traffic.py $O/pmc_c4_f $O/pmc_c4_w > $O/pmc_traffic_c4.json
# keep what is judged: the kernel statistics, the reduced counters, the logs; drop traces and raw counter rows
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*domain_stats.csv" -delete
find $O/pmc_f $O/pmc_w $O/pmc_c4_f $O/pmc_c4_w -name "*.csv" -delete 2>/dev/null
python tools/pmc_issue.py $O/issue > $O/pmc_issue.json
find $O -name "*kernel_stats*" | head
