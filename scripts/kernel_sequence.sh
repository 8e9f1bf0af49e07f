#!/bin/bash
# usage (GPU box): scripts/kernel_sequence.sh <out name> <script.py> [args]  -- rocprofv3 kernel trace of a python script: the LAST
# 400 kernels in launch order with duration and grid size, into gpurun_out/<name>.txt (what runs next to what in a pass)
root=${GRAFT_REPO_ROOT:-$(pwd)}
name=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks
rocprofv3 --kernel-trace --output-format csv -d /tmp/ks -- python3 "$@" > /tmp/ks.log 2>&1 || tail -5 /tmp/ks.log
python3 - > $root/gpurun_out/$name.txt <<'PY'
import csv, glob, re
rows = []
for f in glob.glob("/tmp/ks/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows[-400:]:
    n = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", r["Kernel_Name"])[:90]
    print("%8.2f us  grid %8s x %4s  %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?")), n))
PY
