# kernel averages of the C2 resident loop (both RNG modes) under rocprofv3: bash tools/r6_ns_prof.sh tag
tag=${1:-r6nsprof}; export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/$tag; rm -rf $O; mkdir -p $O
cd /tmp
for mode in pcg64 philox; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o ns_c5_$mode -- python $R/tools/r6_ns_modes.py $mode 64 512 2 > $O/ns_c5_$mode.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o ns_c3 -- python $R/tools/ns_c3.py 16 > $O/ns_c3.log 2>&1
cd $R
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete; find $O -name "*domain_stats.csv" -delete
python - $O <<'PY'
import csv, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/*kernel_stats.csv")):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f.split("/")[-1], "total ms %.1f" % (tot / 1e6))
    for r in rows[:12]:
        print("  %-48s calls %5s avg %8.1f us %5.1f%%" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:48], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
