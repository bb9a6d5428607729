# timeline of the rebuilds of a C3 resident loop (16 runs): bash tools/r6_c3_trace.sh tag [VAR=val ...]
tag=$1; shift
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/$tag; rm -rf $O; mkdir -p $O
cd /tmp
env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python $R/tools/ns_c3.py 16 > $O/run.txt 2>&1
cd $R
python tools/r6_c3_reduce.py $O | tee $O/timeline.txt
find $O -name "*.csv" -delete
