# Round-3 profile collection (run on the GPU box through gpurun): kernel stats of the bench's timed launch
# shape and of the device-resident loops.  usage: bash tools/prof_r03.sh <outdir under gpurun_out>
set -x
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-prof_r03}
mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o bench -- python $R/bench.py --lean --steps 20 --warmup 5 > $O/bench_line_under_rocprof.json 2> $O/rocprof_bench.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o ns_c5 -- python $R/tools/ns_c5.py 64 512 > $O/ns_c5.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o ns_c3 -- python $R/tools/ns_c3.py 16 > $O/ns_c3.log 2>&1
cd $R
find $O -name "*kernel_stats*" | head
