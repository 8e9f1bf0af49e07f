#!/bin/bash
# usage: scripts/ktrace.sh <tag> <prof_target args...>  -> prints per-kernel average ns (rocprofv3 --kernel-trace --stats)
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/kt_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $root/scripts/prof_target.py "$@" > /dev/null 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("   %-44s calls %4s avg %10.1f us" % (r["Name"].split("(anonymous namespace)::")[-1][:44], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
