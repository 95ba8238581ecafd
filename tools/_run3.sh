timeout 900 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_online_decoder.py tests/test_gpu_online_nnet.py -x -q -m gpu 2>&1 | tail -2
KH_FUZZ_SEEDS=100 timeout 900 python -m pytest tests/test_gpu_decoder.py -x -q -m gpu -k "random" 2>&1 | tail -2
BENCH_VERBOSE=1 KH_DECODER_PROFILE=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/b_new.log 2>&1; echo main; grep -E "share of shader|prune by|^\[bench\]|loglike" gpurun_out/b_new.log | tail -4 | cut -c1-420
KH_LIB_OVERRIDE=tools/libkh_exp_prev.so BENCH_VERBOSE=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/b_old.log 2>&1; echo prev; grep -E "^\[bench\]" gpurun_out/b_old.log | tail -1
