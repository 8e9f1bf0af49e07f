#!/bin/bash
# usage (GPU box): scripts/big_kernels.sh [min_us]   -> kernels of a bench step that take >= min_us, grouped by (name, grid)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/kt_big
min=${1:-25}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-micro > /dev/null 2>&1
python3 - "$out" "$min" <<'PY'
import csv, glob, sys, re, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
minus = float(sys.argv[2])
steps = 13.0
agg = collections.defaultdict(lambda: [0, 0.0])
tot = 0.0
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    if d >= minus:
        name = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", r["Kernel_Name"])[:90]
        key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "?")))
        agg[key][0] += 1; agg[key][1] += d
print("total kernel time per step %.2f ms; kernels >= %.0f us:" % (tot / steps / 1e3, minus))
for (name, grid), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print("%7.3f ms/step  x%5.1f/step  avg %7.1f us  grid %-9s %s" % (t / steps / 1e3, n / steps, t / n, grid, name))
PY
