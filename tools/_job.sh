python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 900 python tools/validate_bench_scale.py 700 2>&1 | tail -6
KH_FUZZ_SEEDS=400 timeout 1200 python -m pytest tests/test_gpu_decoder.py -x -q -k fuzz 2>&1 | tail -3
