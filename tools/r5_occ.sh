# the generator pass at several grids (blocks per CU): bash tools/r5_occ.sh
R=$PWD
mkdir -p gpurun_out/r5occ
for b in 0 8 6 5 4; do
  if [ $b = 0 ]; then unset DH_ITEMGEN_BLOCKS_PER_CU; else export DH_ITEMGEN_BLOCKS_PER_CU=$b; fi
  echo "blocks_per_cu=$b $(timeout 200 python tools/rng_launch_prof.py 60)" | tee -a gpurun_out/r5occ/occ.txt
done
unset DH_ITEMGEN_BLOCKS_PER_CU
bash tools/r5_rng.sh r5occ
timeout 600 python -m pytest tests/test_gpu_rwalkq.py -x -q 2>&1 | tail -3
