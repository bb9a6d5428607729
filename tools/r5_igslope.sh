# marginal cost of sixteen extra instructions of one kind per generator round: builds variants/libdynhip_x<k>.so with
# -DDH_IG_EXTRA=k (1 s_add, 2 v_add_u32, 3 v_mad_u64_u32, 4 ds_bpermute, 5 s_nop, 6 s_branch, 7 v_readlane) HERE
# (bash tools/r5_igslope.sh build), then on the GPU box: bash tools/r5_igslope.sh
if [ "$1" = build ]; then
  mkdir -p variants /tmp/wq
  cd dynesty_amd/csrc
  for k in 1 2 3 4 5 6 7; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DDH_IG_EXTRA=$k -c walkq.hip -o /tmp/wq/walkq_x$k.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls *.o | grep -v "walkq.o\|_timing\|_ablate\|_prof\|asan") /tmp/wq/walkq_x$k.o -o ../../variants/libdynhip_x$k.so
  done
  exit 0
fi
mkdir -p gpurun_out/r5slope
echo "base $(timeout 200 python tools/rng_launch_prof.py 60)" | tee gpurun_out/r5slope/slope.txt
for k in 1 2 3 4 5 6 7; do
  echo "x$k $(DYNHIP_LIB=$PWD/variants/libdynhip_x$k.so timeout 200 python tools/rng_launch_prof.py 60)" | tee -a gpurun_out/r5slope/slope.txt
done
echo "base $(timeout 200 python tools/rng_launch_prof.py 60)" | tee -a gpurun_out/r5slope/slope.txt
