# A/B of walk-kernel builds on one box: per-kernel averages of a lean bench under rocprofv3 (names inside dynesty_amd/)
#   bash tools/r6_walk_ab.sh <tag> lib1.so lib2.so ...
cd /root/repo; R=$PWD; O=$R/gpurun_out/${1:-r6walkab}; mkdir -p $O; shift
export TMPDIR=/tmp
for rep in 1 2; do
for lib in "$@"; do
  D=$O/prof_${lib%.so}_$rep; rm -rf $D
  (cd /tmp && DYNHIP_LIB=$R/dynesty_amd/$lib timeout 600 rocprofv3 --kernel-trace --stats -d $D -o t --output-format csv -- python $R/bench.py --lean --steps 20 --warmup 5 > $D.log 2>&1)
  python - "$D" "$lib" <<'PY' | tee -a $O/ab.txt
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = {r["Name"]: r for r in csv.DictReader(open(f))}
out = []
for k, r in rows.items():
    if "rwalkq_kernel" in k or "itemgen_kernel" in k:
        out.append("%s %.1f us x %s" % (k.split("(")[0].replace("(anonymous namespace)::", "").replace("void ", "")[:40], float(r["AverageNs"]) / 1e3, r["Calls"]))
print(sys.argv[2], " | ".join(out))
PY
  find $D -name "*.csv" -delete
done
done
