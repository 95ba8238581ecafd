#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of DecodeKernel for one bench step (two separate passes); run on the GPU box:
#   tools/pmc_traffic.sh OUTDIR
out=${1:-gpurun_out/traffic}; mkdir -p "$out"; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d "$out/pmc_$c" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-strong > "$out/pmc_$c.json" 2> "$out/pmc_$c.log"
  f=$(find "$out/pmc_$c" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 tools/pmc_summarize.py "$f" DecodeKernel | tee "$out/$c.txt"
  rm -rf "$out/pmc_$c"
done
