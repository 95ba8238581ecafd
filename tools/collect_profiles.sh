#!/bin/bash
# Round profile bundle; run on the GPU box from the repo root:
#   tools/collect_profiles.sh gpurun_out/profiles_rNN rNN
# 1. rocprofv3 --kernel-trace --stats of the bench command
# 2. PMC passes (separate runs): FETCH_SIZE, WRITE_SIZE of DecodeKernel (the library's default: reference order)
# 3. bench line (default workload) + decoder phase shares, with the traffic of (2)
out=${1:-gpurun_out/profiles}; tag=${2:-r01}
mkdir -p "$out"
export TMPDIR=/tmp
T=${PROFILE_TIMEOUT:-420}
timeout -k 10 $T rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-extra-legs > "$out/${tag}_bench_under_rocprof.json" 2> "$out/kt.log"
f=$(find "$out/kt" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/${tag}_bench_kernel_stats.csv"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 $T rocprofv3 --pmc $c --output-format csv -d "$out/pmc_$c" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-extra-legs > "$out/pmc_$c.json" 2> "$out/pmc_$c.log"
  f=$(find "$out/pmc_$c" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 tools/pmc_summarize.py "$f" DecodeKernel > "$out/${tag}_pmc_$c.txt"
done
# request sizes and DRAM-side byte tallies of DecodeKernel (the FETCH_SIZE correction, see tools/pmc_calibrate.sh)
{
  echo "# DecodeKernel, one launch: read request sizes and DRAM-side tallies in 32-byte units (separate rocprofv3 --pmc passes)"
  for set in "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_WRREQ_WRITE_DRAM_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum"; do
    rm -rf "$out/pmc_x"
    timeout -k 10 $T rocprofv3 --pmc $set --output-format csv -d "$out/pmc_x" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-extra-legs --no-strong > /dev/null 2> "$out/pmc_x.log"
    f=$(find "$out/pmc_x" -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python3 tools/pmc_summarize.py "$f" DecodeKernel
  done
  rm -rf "$out/pmc_x"
} > "$out/${tag}_pmc_request_sizes.txt"
# issue / wait split of DecodeKernel (the "HBM-bound" label is half the story): VALU instructions and the
# cycles waves spend issuing them / issuing vector-memory instructions / waiting
rm -rf "$out/pmc_sq"
timeout -k 10 $T rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$out/pmc_sq" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-extra-legs --no-end-to-end > /dev/null 2> "$out/pmc_sq.log"
f=$(find "$out/pmc_sq" -name "*counter_collection.csv" | head -1)
{ echo "# DecodeKernel<reference order> (the default), rocprofv3 --pmc pass over bench.py --steps 1 (warm-up 0): SQ counters summed over the run's DecodeKernel dispatches"; [ -n "$f" ] && python3 tools/pmc_summarize.py "$f" DecodeKernel; } > "$out/${tag}_pmc_sq.txt"
rm -rf "$out/pmc_sq"
# a15: kernel trace of the lattice forward-backward leg (256 and 2048 lattices: tools/bench_lattice_fb.py)
rm -rf "$out/kt_lat"
timeout -k 10 $T rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt_lat" -- python3 tools/bench_lattice_fb.py > "$out/${tag}_lattice_fb.json" 2> "$out/kt_lat.log"
f=$(find "$out/kt_lat" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/${tag}_lattice_fb_kernel_stats.csv"
rm -rf "$out/kt_lat"
# the opt-in order-independent rule (the headline kernel of rounds 1-5; round 6's default is the reference's own order,
# which every pass above measured): phase shares
KH_DECODER_ORDER=canonical KH_DECODER_PROFILE=1 timeout $T python3 bench.py --steps 1 --warmup 0 --no-secondary --no-extra-legs --no-end-to-end --no-cpu-baseline > /dev/null 2> "$out/canon.err"
python3 tools/phases_extract.py "$out/canon.err" > "$out/${tag}_decoder_phases_canonical.txt"
# ... and its bytes and instructions (value_canonical / roofline.canonical_* of the bench line)
{
  echo "# DecodeKernel<canonical> (KH_DECODER_ORDER=canonical), one launch: separate rocprofv3 --pmc passes over bench.py --steps 1 --warmup 0"
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
    rm -rf "$out/pmc_x"
    KH_DECODER_ORDER=canonical timeout -k 10 $T rocprofv3 --pmc $set --output-format csv -d "$out/pmc_x" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-extra-legs --no-end-to-end > /dev/null 2> "$out/pmc_x.log"
    f=$(find "$out/pmc_x" -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python3 tools/pmc_summarize.py "$f" DecodeKernel | grep -v "^#"
  done
  rm -rf "$out/pmc_x"
} > "$out/${tag}_pmc_canonical.txt"
# the bench line LAST: its roofline.traffic reads the PMC record of THIS build (pmc_record.py fails when a pass
# left no summary, and stamps the record with the kernel source's hash: bench.py refuses a record of another build)
python3 tools/pmc_record.py "$out/${tag}_pmc_FETCH_SIZE.txt" "$out/${tag}_pmc_WRITE_SIZE.txt" "$out/${tag}_bench_kernel_stats.csv" "$out/${tag}_pmc_traffic.json" reference-order || { echo "collect_profiles: PMC record incomplete" >&2; exit 1; }
cp "$out/${tag}_pmc_FETCH_SIZE.txt" "$out/${tag}_pmc_WRITE_SIZE.txt" "$out/${tag}_pmc_traffic.json" profiles/ || exit 1
# (every leg of the line, the CPU baselines included: 5-8 minutes depending on the box - its own, longer limit)
KH_DECODER_PROFILE=1 BENCH_VERBOSE=1 timeout ${PROFILE_BENCH_TIMEOUT:-1200} python3 bench.py --steps 3 --warmup 1 > "$out/${tag}_bench.json" 2> "$out/bench.err"
[ -s "$out/${tag}_bench.json" ] || { echo "collect_profiles: the bench line is empty (timeout?)" >&2; exit 1; }
python3 tools/phases_extract.py "$out/bench.err" > "$out/${tag}_decoder_phases.txt"
rm -rf "$out/kt" "$out"/pmc_FETCH_SIZE "$out"/pmc_WRITE_SIZE
ls -la "$out"
