O=gpurun_out/${1:-r5fx}; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_resident_mirror.py tests/test_gpu_ns_ensemble.py -x -q 2>&1 | tail -15 | tee $O/pytest.txt
timeout 900 python tools/forced_exact_cmp.py $O/forms.jsonl 2>&1 | tail -30
