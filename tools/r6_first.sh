# round 6, first call on the box (run through tools/stage_reference.sh): same-seed whole runs on the HIP backend,
# the reference's Pool on the host cores, phase split + timeline + A/B time of the rebuild, a bench line
O=gpurun_out/r6first; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_same_seed_hw.py -q -x 2>&1 | tail -15 | tee $O/same_seed.txt
cp gpurun_out/same_seed_hw.json $O/ 2>/dev/null
timeout 120 python tools/rb_ab5.py 60 2>&1 | tee $O/rb_ab.txt
timeout 200 python tools/rb_phase_batch.py 1 64 2>&1 | tee $O/rb_phase.txt
bash tools/r5_trace.sh r6first_tl64 64 > $O/timeline64.txt 2>&1
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
timeout 600 python tools/ref_pool_hw.py $O 32,128,256 30 2>&1 | tail -40 > $O/refpool.txt
