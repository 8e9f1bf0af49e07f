#!/bin/bash
# usage: scripts/ktrace_grid.sh <tag> <script> [args]: rocprofv3 kernel trace; msda kernels grouped by grid size
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/ktg_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $root/"$@" > $out.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, re, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(msda_\w+(<[^>]*>)?)", r["Kernel_Name"])
        if m:
            agg[(m.group(1), r["Grid_Size_X"], r["Workgroup_Size_X"], r["LDS_Block_Size"], r["VGPR_Count"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items()):
    v.sort()
    print("%-26s grid %8s wg %5s lds %7s vgpr %4s : n=%4d median %9.1f us  min %9.1f" % (k + (len(v), v[len(v) // 2], v[0])))
PY
