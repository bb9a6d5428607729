O=gpurun_out/r6second; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_same_seed_hw.py -q 2>&1 | tail -15 | tee $O/same_seed.txt
cp gpurun_out/same_seed_hw.json $O/ 2>/dev/null
bash tools/r5_trace.sh r6second_tl64 64 > $O/timeline64.txt 2>&1
bash tools/r5_trace.sh r6second_tl1 1 > $O/timeline1.txt 2>&1
