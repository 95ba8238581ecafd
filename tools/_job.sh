python -m pytest tests -m gpu -x -q 2>&1 | tail -3
tools/collect_profiles.sh gpurun_out/profiles_r02m r02 > gpurun_out/collect.log 2>&1
tail -2 gpurun_out/collect.log
cat gpurun_out/profiles_r02m/r02_pmc_request_sizes.txt gpurun_out/profiles_r02m/r02_pmc_FETCH_SIZE.txt gpurun_out/profiles_r02m/r02_pmc_WRITE_SIZE.txt
