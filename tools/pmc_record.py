"""Assemble profiles/<tag>_pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE summaries of tools/collect_profiles.sh.

usage: pmc_record.py FETCH_SUMMARY WRITE_SUMMARY KERNEL_STATS_CSV OUT_JSON [reference-order|canonical]
Fails (exit 1) when a summary is missing or empty, so that a timed-out PMC pass can never leave an older
figure standing as the current one.  The record carries the hash of the kernel's source file: bench.py quotes
the traffic only while that hash matches the library it runs."""
import csv
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SRC = os.path.join(ROOT, "old-kaldi-git_amd", "csrc", "kh_decoder.hip")


def kernel_src_sha16():
    with open(KERNEL_SRC, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def counter(path, name):
    """Per-LAUNCH value of a counter: tools/pmc_summarize.py prints `name,<n> dispatches,<sum over them>` (a bench run
    launches DecodeKernel once per step of each timed loop; the launches of one run do the same work)."""
    with open(path) as f:
        for line in f:
            fields = line.strip().split(",")
            if len(fields) >= 3 and fields[0] == name:
                n = int(fields[1].split()[0]) if fields[1].split() and fields[1].split()[0].isdigit() else 1
                return float(fields[2]) / max(n, 1), n
    raise SystemExit("pmc_record: %s has no %s row" % (path, name))


def kernel_ms(stats_csv, pat="DecodeKernel"):
    """Average duration, launches and the search of the run's DecodeKernel<lazy schedule, reference order> instantiation
    (a run of bench.py --no-extra-legs launches exactly one of them)."""
    rows = []
    with open(stats_csv) as f:
        for r in csv.DictReader(f):
            if pat in r.get("Name", ""):
                rows.append(r)
    if not rows:
        raise SystemExit("pmc_record: %s has no %s row" % (stats_csv, pat))
    if len(rows) > 1:
        raise SystemExit("pmc_record: %s holds %d %s instantiations (run bench.py with --no-extra-legs)" % (stats_csv, len(rows), pat))
    r = rows[0]
    exact = "true>" in r["Name"].split(pat, 1)[1].split("(", 1)[0].replace(" ", "")
    return float(r["AverageNs"]) / 1e6, int(r["Calls"]), ("reference-order" if exact else "canonical")


def main():
    fetch, write, stats, out = sys.argv[1:5]
    want = sys.argv[5] if len(sys.argv) > 5 else None      # the search the passes were meant to time (a mislabelled record fails)
    for p in (fetch, write, stats):
        if not os.path.exists(p) or os.path.getsize(p) == 0:
            raise SystemExit("pmc_record: %s is missing or empty" % p)
    ms, calls, search = kernel_ms(stats)
    if want is not None and want != search:
        raise SystemExit("pmc_record: the kernel trace holds the %s kernel, the record was to be the %s one" % (search, want))
    (fs, fn), (ws, wn) = counter(fetch, "FETCH_SIZE"), counter(write, "WRITE_SIZE")
    rec = {"kernel": "DecodeKernel<%s>" % ("reference order" if search == "reference-order" else "canonical"), "search": search,
           "kernel_src_sha16": kernel_src_sha16(),
           "FETCH_SIZE_KiB": fs, "WRITE_SIZE_KiB": ws, "pmc_dispatches": [fn, wn],
           "kernel_trace_avg_ms": ms, "kernel_trace_calls": calls,
           "note": "rocprofv3 --pmc, per launch (the pass total / its dispatches); one launch = 2620 utterances / 1.94 M frames; traffic = 2 x FETCH_SIZE + WRITE_SIZE "
                   "(read correction: profiles/r02_pmc_calibration.txt)"}
    with open(out, "w") as f:
        json.dump(rec, f, indent=1)
        f.write("\n")
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
