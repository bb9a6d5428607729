# final records of round 6 on the box (through tools/stage_reference.sh): the bench line with the real reference as CPU
# baseline, tap B, the rebuild timelines, the nbound conventions side by side
O=gpurun_out/r6final; mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line_with_reference.json 2> $O/bench.err
timeout 900 python tools/tapb_hw.py $O 30 > $O/tapb.log 2>&1
bash tools/r5_trace.sh r6final_tl64 64 > $O/rebuild_timeline_64runs.txt 2>&1
bash tools/r5_trace.sh r6final_tl1 1 > $O/rebuild_timeline_1run.txt 2>&1
bash tools/r5_trace.sh r6final_tl128 128 > $O/rebuild_timeline_128runs.txt 2>&1
timeout 900 python tools/forced_exact_cmp.py $O/forced_exact_forms.jsonl > $O/forced.log 2>&1
python tools/r6_phase.py 1 64 > $O/rebuild_phases.txt 2>&1
python tools/r6_wgclock.py 64 > $O/wgclock_64runs.txt 2>&1
