#!/bin/bash
# PMC passes over one bench step (DecodeKernel is the row of interest); run on the GPU box:
#   tools/pmc_profile.sh OUTDIR
# Counters are collected in separate passes (TCC has 4 slots: FETCH_SIZE takes 3, WRITE_SIZE 2).
out=${1:-gpurun_out/pmc}
mkdir -p "$out"
export TMPDIR=/tmp
run() {
  name=$1; shift
  timeout -k 10 ${PMC_TIMEOUT:-300} rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary ${PMC_BENCH_ARGS:-} > "$out/$name.log" 2>&1
  f=$(find "$out/$name" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 tools/pmc_summarize.py "$f" DecodeKernel > "$out/$name.summary.txt"
  cat "$out/$name.summary.txt"
}
if [ -n "$PMC_ONLY" ]; then run custom $PMC_ONLY; exit 0; fi
run sq SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
# (the TA_* and TCP_* counters abort rocprofv3 on this pool: not collected)
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_ATOMIC_sum
