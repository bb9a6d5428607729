# durations of the work-queue tail (k_tree) over a C2 resident loop: bash tools/r6_ktree.sh tag [VAR=val ...]
tag=$1; shift
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/$tag; rm -rf $O; mkdir -p $O
cd /tmp
env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python $R/tools/r6_ns_modes.py pcg64 64 512 1 > $O/run.txt 2>&1
cd $R
python - $O <<'PY' | tee $O/ktree.txt
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
roots = [i for i, r in enumerate(rows) if "k_root_parts" in r["Kernel_Name"]]
for n, i in enumerate(roots):
    j = roots[n + 1] if n + 1 < len(roots) else len(rows)
    seg = rows[i:j]
    fin = [k for k, r in enumerate(seg) if "k_out_eig" in r["Kernel_Name"]]
    end = seg[fin[0]] if fin else seg[-1]
    tot = (int(end["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
    kt = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in seg if "k_tree" in r["Kernel_Name"]]
    kf = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in seg if "k_finish" in r["Kernel_Name"]]
    print("rebuild %2d: %7.1f us root..k_out_eig, k_tree %7.1f, k_finish %6.1f" % (n, tot, sum(kt), sum(kf)))
PY
python tools/r6_first_rebuild.py $O | tee $O/first_rebuild.txt
find $O -name "*.csv" -delete
