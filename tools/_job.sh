tools/collect_profiles.sh gpurun_out/profiles_r02i r02 > gpurun_out/collect.log 2>&1
tail -3 gpurun_out/collect.log
cat gpurun_out/profiles_r02i/r02_pmc_request_sizes.txt gpurun_out/profiles_r02i/r02_pmc_FETCH_SIZE.txt gpurun_out/profiles_r02i/r02_pmc_WRITE_SIZE.txt
tail -c 1500 gpurun_out/profiles_r02i/r02_bench.json
