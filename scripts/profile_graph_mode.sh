#!/bin/bash
# GPU box: kernel-trace stats of bench.py with the opt-in transformer hipGraph path -> gpurun_out/profiles_<tag>/bench_graph_kernel_stats.txt
root=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r01}
out=$root/gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_graph_trace -- python3 $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --transformer-graph > $out/bench_graph_under_rocprof.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, re
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/bench_graph_trace/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(out + "/bench_graph_kernel_stats.txt", "w") as fh:
    fh.write("rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --transformer-graph\n")
    fh.write("total kernel time %.1f ms over %d kernel names (13 steps incl. warm-up; the warm-up also holds the 3 eager passes and the capture pass of every graphed piece)\n" % (tot / 1e6, len(rows)))
    fh.write("%7s %11s %8s %12s  %s\n" % ("share", "total_ms", "calls", "avg_us", "kernel"))
    for r in rows[:40]:
        name = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", r["Name"])[:110]
        fh.write("%6.2f%% %11.3f %8s %12.2f  %s\n" % (100 * float(r["TotalDurationNs"]) / tot, float(r["TotalDurationNs"]) / 1e6, r["Calls"], float(r["AverageNs"]) / 1e3, name))
print(open(out + "/bench_graph_kernel_stats.txt").read()[:1500])
tail = open(out + "/bench_graph_under_rocprof.log").read().strip().splitlines()[-1]
print(tail[:300])
PY
