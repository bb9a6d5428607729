# final records of round 6, second half, on the box (through tools/stage_reference.sh)
O=gpurun_out/r6final2; mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line_with_reference.json 2> $O/bench.err
timeout 900 python -m pytest tests/test_gpu_same_seed_hw.py -q 2>&1 | tail -5 > $O/same_seed.txt
cp gpurun_out/same_seed_hw.json $O/ 2>/dev/null
timeout 900 python tools/tapb_hw.py $O 30 > $O/tapb.log 2>&1
bash tools/r5_trace.sh r6final2_tl64 64 > $O/rebuild_timeline_64runs.txt 2>&1
bash tools/r5_trace.sh r6final2_tl1 1 > $O/rebuild_timeline_1run.txt 2>&1
bash tools/r5_trace.sh r6final2_tl128 128 > $O/rebuild_timeline_128runs.txt 2>&1
python tools/r6_rb.py 30 1 16 64 128 256 > $O/rebuild_ms.json 2>&1
bash tools/r6_c3_trace.sh r6final2_c3 > /dev/null 2>&1; cp gpurun_out/r6final2_c3/timeline.txt $O/c3_rebuild_timeline.txt
bash tools/r6_ktree.sh r6final2_kt > /dev/null 2>&1; cp gpurun_out/r6final2_kt/ktree.txt $O/c2_loop_rebuild_durations.txt
timeout 900 python tools/forced_exact_cmp.py $O/forced_exact_forms.jsonl > $O/forced.log 2>&1
bash tools/r5_det.sh > $O/determinism.txt 2>&1
