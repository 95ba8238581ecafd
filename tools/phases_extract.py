"""Picks the decoder's phase report of the bench's main launch (the last 2620-utterance / largest
launch) out of the stderr of `KH_DECODER_PROFILE=1 BENCH_VERBOSE=1 python bench.py` (the secondary
legs decode too and print their own reports).   python tools/phases_extract.py STDERR_FILE > OUT"""
import re
import sys

lines = [l for l in open(sys.argv[1]) if l.startswith("[kh_decoder profile]") or l.startswith("[bench]")]
launch = [(i, int(re.search(r"launch 0: (\d+) utterances", l).group(1))) for i, l in enumerate(lines) if "launch 0:" in l]
if not launch:
    sys.exit("no decoder profile in " + sys.argv[1])
big = max(n for _, n in launch)
i = [i for i, n in launch if n == big][-1]
j = i
while j > 0 and "host:" in lines[j - 1]:
    j -= 1
k = i + 1
while k < len(lines) and "launch 0" not in lines[k] and "arenas" not in lines[k] and not lines[k].startswith("[bench]"):
    k += 1
arenas = [l for l in lines[:i] if "arenas" in l][-1:]
print("# KH_DECODER_PROFILE=1 BENCH_VERBOSE=1 python bench.py (last timed step of the main workload; shader-cycle shares "
      "from s_memtime stamps of thread 0 of every workgroup)")
sys.stdout.write("".join(arenas + lines[j:k] + [l for l in lines if l.startswith("[bench]")][-1:]))
