tools/collect_profiles.sh gpurun_out/profiles_r02j r02 > gpurun_out/collect.log 2>&1
tail -2 gpurun_out/collect.log
cat gpurun_out/profiles_r02j/r02_pmc_request_sizes.txt gpurun_out/profiles_r02j/r02_pmc_FETCH_SIZE.txt gpurun_out/profiles_r02j/r02_pmc_WRITE_SIZE.txt
head -c 600 gpurun_out/profiles_r02j/r02_bench.json
