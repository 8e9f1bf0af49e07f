#!/bin/bash
# usage (GPU box): scripts/small_kernels.sh [max_us] [extra bench args]  -> kernels of a bench step below max_us, grouped by (name, grid)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=/tmp/kt_small
max=${1:-20}; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $out
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-micro "$@" > /dev/null 2>&1
python3 - "$out" "$max" <<'PY'
import csv, glob, sys, re, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
mx = float(sys.argv[2])
steps = 13.0
agg = collections.defaultdict(lambda: [0, 0.0])
tot = small = 0.0; nsmall = 0
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    if d < mx:
        small += d; nsmall += 1
        name = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", r["Kernel_Name"])
        name = re.sub(r"<.*", "", name)[:60] + " | " + (re.search(r"(\w+Functor|\w+_kernel_cuda|\w+_kernel_impl|Cijk_\w{8})", r["Kernel_Name"]) or [""])[0][:40]
        key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", "?"))
        agg[key][0] += 1; agg[key][1] += d
print("total kernel time per step %.2f ms; kernels < %.0f us: %.2f ms/step in %.0f launches/step" % (tot / steps / 1e3, mx, small / steps / 1e3, nsmall / steps))
for (name, grid, wg), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:90]:
    print("%7.3f ms/step  x%6.1f/step  avg %6.2f us  grid %-9s wg %-4s %s" % (t / steps / 1e3, n / steps, t / n, grid, wg, name))
PY
