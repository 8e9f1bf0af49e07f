# usage (GPU box): scripts/kstats_py.sh <out name> <script.py> [args]  -- rocprofv3 kernel-trace stats of a python script, top 45 kernels
root=${GRAFT_REPO_ROOT:-$(pwd)}
name=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -- python3 "$@" > /tmp/kp.log 2>&1 || tail -5 /tmp/kp.log
python3 - > $root/gpurun_out/$name.txt <<'PY'
import csv, glob, re
rows = []
for f in glob.glob("/tmp/kp/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.2f ms over %d launches" % (tot / 1e6, sum(int(r["Calls"]) for r in rows)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:45]:
    print("%6.2f%% %8.2f ms calls %6s avg %8.2f us  %s" % (100 * float(r["TotalDurationNs"]) / tot, float(r["TotalDurationNs"]) / 1e6, r["Calls"], float(r["AverageNs"]) / 1e3, re.sub(r"\s+", " ", r["Name"])[:150]))
PY
