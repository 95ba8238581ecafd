#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration in the decoder's access shapes (tools/pmc_calibrate.hip); on the GPU box:
#   tools/pmc_calibrate.sh OUTDIR
out=${1:-gpurun_out/calib}; mkdir -p "$out"; export TMPDIR=/tmp
exe=old-kaldi-git_amd/build/pmc_calibrate
[ -x $exe ] || hipcc --offload-arch=gfx950 -O2 -o $exe tools/pmc_calibrate.hip || exit 1
$exe > "$out/known_bytes.txt"; cat "$out/known_bytes.txt"
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_[A-Z0-9_]*\|TCC_[A-Z0-9_]*DRAM[A-Z0-9_]*\|FETCH_SIZE\|WRITE_SIZE" | sort -u > "$out/counters_available.txt"
pass() {
  name=$1; shift
  timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- $exe > "$out/$name.log" 2>&1
  f=$(find "$out/$name" -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python3 - "$f" > "$out/$name.txt" <<'PY'
import csv, sys
from collections import OrderedDict
rows = OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    rows.setdefault(k, {})[r["Counter_Name"]] = rows.get(k, {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k, v in rows.items():
    if k.startswith("Cal"):
        print(k, " ".join("%s=%.6g" % kv for kv in sorted(v.items())))
PY
    cat "$out/$name.txt"
  else
    tail -3 "$out/$name.log"
  fi
  rm -rf "$out/$name"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
pass wrreq TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
pass atomic TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum
# request sizes and the DRAM-side byte tallies (32-B units)
pass rdsize TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum
pass rddram TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_32B_sum
pass wrdram TCC_EA0_WRREQ_WRITE_DRAM_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum
pass atdram TCC_EA0_WRREQ_ATOMIC_DRAM_sum TCC_EA0_WRREQ_ATOMIC_DRAM_32B_sum
