"""Sum rocprofv3 counter_collection.csv rows per counter for kernels matching a name."""
import csv
import sys
from collections import defaultdict

path, pat = sys.argv[1], sys.argv[2]
tot, n = defaultdict(float), defaultdict(int)
with open(path) as f:
    for r in csv.DictReader(f):
        if pat not in r.get("Kernel_Name", ""):
            continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
        n[r["Counter_Name"]] += 1
for k in sorted(tot):
    print("%s,%d dispatches,%.6g" % (k, n[k], tot[k]))
