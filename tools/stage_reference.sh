#!/bin/bash
# The real reference beside the benchmark on an MI355X box.  The reference tree does not
# travel with the repository (and its sources are never committed): stage a scratch copy, run, delete.
#
#   bash tools/stage_reference.sh 'python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_with_reference.json'
#
# bench.py's cpu_baseline then times dynesty itself (kind "reference"); without the copy it times the oracle port
# (kind "port").  tools/tapb_hw.py (tap B: the unmodified NestedSampler over the drop-in classes) uses the same copy.
set -e
cd "$(dirname "$0")/.."
mkdir -p _refstage && cp -r /root/reference/py _refstage/py
trap 'rm -rf _refstage' EXIT
/usr/local/graft/bin/gpurun --timeout ${TIMEOUT:-1500} -- "export DYNESTY_REF_PY=\$PWD/_refstage/py; $1"
