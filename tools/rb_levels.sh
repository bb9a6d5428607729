export TMPDIR=/tmp
R=$PWD
cd /tmp
for n in ${RUNS:-1 4 64}; do
  rm -rf $R/gpurun_out/rbl_$n
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/rbl_$n -o t -- python $R/tools/rb_levels.py $n > /dev/null 2>&1
  echo "=== runs=$n"
  python $R/tools/rb_trace.py $(find $R/gpurun_out/rbl_$n -name "*kernel_trace.csv" | head -1) -1
done
