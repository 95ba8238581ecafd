python -m pytest tests/test_gpu_decoder.py tests/test_gpu_structured.py tests/test_gpu_online_decoder.py -x -q 2>&1 | tail -2
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-strong 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])"
bash tools/pmc_traffic.sh gpurun_out/traffic_clear 2>&1 | tail -2
