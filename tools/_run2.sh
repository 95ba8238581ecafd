set -x
timeout 900 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_online_decoder.py tests/test_gpu_online_nnet.py -x -q -m gpu 2>&1 | tail -3
BENCH_VERBOSE=1 KH_DECODER_PROFILE=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/b_new.log 2>&1; grep -E "share of shader|^\[bench\]|loglike_per_frame" gpurun_out/b_new.log | tail -4 | cut -c1-420
KH_LIB_OVERRIDE=tools/libkh_exp_base.so BENCH_VERBOSE=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/b_old.log 2>&1; grep -E "^\[bench\]" gpurun_out/b_old.log | tail -2
