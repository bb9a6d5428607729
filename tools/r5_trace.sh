# timeline of the last R-run bench rebuild under the given environment settings: bash tools/r5_trace.sh tag R VAR=val ...
tag=$1; R_=$2; shift; shift
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/$tag; rm -rf $O; mkdir -p $O
cd /tmp
env "$@" timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python $R/tools/rb_levels.py $R_ > /dev/null 2>&1
cd $R
python tools/rb_trace_reduce.py $O | tee $O/timeline.txt
find $O -name "*.csv" -delete
