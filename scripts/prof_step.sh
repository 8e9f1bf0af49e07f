#!/bin/bash
# usage: scripts/prof_step.sh <tag> [full_step args]  -> rocprofv3 kernel stats of full training steps
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/step_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $root/scripts/full_step.py "$@" > $out.log 2>&1
tail -4 $out.log
python3 - "$out" <<'PY'
import csv, glob, sys, re
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms over %d kernel names" % (tot / 1e6, len(rows)))
for r in rows[:45]:
    name = re.sub(r"\(anonymous namespace\)::|void |at::native::|\(.*", "", r["Name"])[:78]
    print("%6.2f%% %9.2f ms %7s calls avg %9.1f us  %s" % (100 * float(r["TotalDurationNs"]) / tot, float(r["TotalDurationNs"]) / 1e6, r["Calls"], float(r["AverageNs"]) / 1e3, name))
PY
