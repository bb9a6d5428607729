O=gpurun_out/${1:-r5rng}
mkdir -p $O
R=$PWD
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O -o rng -- python $R/tools/rng_launch_prof.py 40 > $R/$O/rng_launch.json 2> $R/$O/rocprof.log
cd $R
cat $O/rng_launch.json
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
python tools/kstats.py $O/rng_kernel_stats.csv 8
