set -x
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/prof_final
cd /tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -o bench -- python $R/bench.py --lean --steps 20 --warmup 5 > $R/gpurun_out/prof_final/bench_line_under_rocprof.json 2> $R/gpurun_out/prof_final/rocprof.log
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_f -- python $R/bench.py --lean --preroll 0 --steps 3 --warmup 1 > $R/gpurun_out/pmc_f.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_w -- python $R/bench.py --lean --preroll 0 --steps 3 --warmup 1 > $R/gpurun_out/pmc_w.log 2>&1
cd $R
python tools/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w > gpurun_out/prof_final/pmc_traffic.json
cp gpurun_out/prof_final/pmc_traffic.json profiles/r02/pmc_traffic.json
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/prof_final/bench_line_final.json 2> gpurun_out/prof_final/bench_final.err
tail -c 600 gpurun_out/prof_final/bench_line_final.json
find gpurun_out/prof_final -name "*stats*" | head
