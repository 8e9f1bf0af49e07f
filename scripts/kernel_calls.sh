# usage (GPU box): scripts/kernel_calls.sh <kernel regex> <script.py> [args]  -- every call of the matching kernels in launch order with its
# duration and grid (rocprofv3 kernel trace), for kernels whose cost differs from call to call (the Swin stages, the encoder levels)
root=${GRAFT_REPO_ROOT:-$(pwd)}
pat=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kc
rocprofv3 --kernel-trace --output-format csv -d /tmp/kc -- python3 "$@" > /tmp/kc.log 2>&1 || tail -5 /tmp/kc.log
python3 - "$pat" <<'PY'
import csv, glob, re, sys
pat = re.compile(sys.argv[1])
rows = []
for f in glob.glob("/tmp/kc/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat.search(r["Kernel_Name"]):
            rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r["Kernel_Name"][:60]))
rows.sort()
for i, (_, d, g, n) in enumerate(rows[-int(len(rows) / max(1, len(rows) // 40)):] if len(rows) > 80 else rows):
    print("%3d %9.1f us  grid %-10s %s" % (i, d, g, n))
PY
