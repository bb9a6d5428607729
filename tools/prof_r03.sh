# Round-3 profile collection (run on the GPU box through gpurun): kernel stats of the bench's timed launch
# shape and of the device-resident loops, and the two PMC passes behind profiles/r03/pmc_traffic.json.
# usage: bash tools/prof_r03.sh <outdir under gpurun_out>
set -x
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${1:-prof_r03}
mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o bench -- python $R/bench.py --lean --steps 20 --warmup 5 > $O/bench_line_under_rocprof.json 2> $O/rocprof_bench.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o ns_c5 -- python $R/tools/ns_c5.py 64 512 > $O/ns_c5.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o ns_c3 -- python $R/tools/ns_c3.py 16 > $O/ns_c3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o ns_c4 -- python $R/tools/c4_ksweep.py 16 128 > $O/ns_c4.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o ns_c1 -- python $R/tools/ns_c1.py 64 64 > $O/ns_c1.log 2>&1
# HBM traffic: separate --pmc passes, --kernel-trace only (MI355X_MICROARCH.md)
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -- python $R/bench.py --lean --preroll 0 --steps 3 --warmup 1 > $O/pmc_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -- python $R/bench.py --lean --preroll 0 --steps 3 --warmup 1 > $O/pmc_w.log 2>&1
cd $R
python tools/pmc_traffic.py $O/pmc_f $O/pmc_w > $O/pmc_traffic.json
find $O -name "*kernel_stats*" | head
