# the resident loop in both RNG modes, timed and under rocprofv3 --kernel-trace --stats
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r6philox; rm -rf $O; mkdir -p $O
for m in pcg64 philox; do
  python tools/r6_ns_modes.py $m 64 512 3 2>&1 | tee $O/time_$m.txt
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$m -o p -- python $R/tools/r6_ns_modes.py $m 64 512 1 > /dev/null 2>&1)
  f=$(find $O/prof_$m -name "*kernel_stats.csv" | head -1)
  cp $f $O/kernel_stats_$m.csv
  find $O/prof_$m -type f -delete
  echo "--- $m top kernels"; python - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats_$m.csv")))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total device ms", round(tot/1e6,2))
for r in rows[:14]:
    print(f'{float(r["TotalDurationNs"])/1e6:8.2f} ms {int(r["Calls"]):6d} x {float(r["AverageNs"])/1e3:8.1f} us  {r["Name"][:90]}')
PY
done
