#!/usr/bin/env python3
"""Dev: bucket a profiles/*bench_kernel_stats.txt-style table (or the rocprof stats csv) by kernel family / size."""
import re, sys, collections
steps = 13.0
rows = []
for l in open(sys.argv[1]):
    m = re.match(r"\s*([\d.]+)%\s+([\d.]+)\s+(\d+)\s+([\d.]+)\s+(.*)", l)
    if m:
        rows.append((float(m.group(2)), int(m.group(3)), float(m.group(4)), m.group(5)))
    if l.startswith("hand-written"):
        break
fam = collections.defaultdict(lambda: [0.0, 0])
for tot, calls, avg, name in rows:
    if name.startswith("Cijk"):
        k = "GEMM (Tensile) avg>=100us" if avg >= 100 else ("GEMM 20-100us" if avg >= 20 else "GEMM <20us")
    elif re.match(r"msda_", name):
        k = "msda"
    elif re.match(r"(ln_|bis_|xty_|rsb_)", name):
        k = "own other"
    elif "elementwise" in name or "Functor" in name or "copyBuffer" in name or "fill" in name.lower():
        k = "elementwise avg>=15us" if avg >= 15 else "elementwise <15us"
    else:
        k = "other avg>=15us" if avg >= 15 else "other <15us"
    fam[k][0] += tot; fam[k][1] += calls
tot = sum(v[0] for v in fam.values())
print("listed rows cover %.2f ms/step" % (tot / steps))
for k, (t, c) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
    print("%-28s %7.2f ms/step  %7.1f launches/step" % (k, t / steps, c / steps))
