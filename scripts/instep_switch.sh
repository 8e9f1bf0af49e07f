# usage (GPU box): scripts/instep_switch.sh NAME=V [NAME=V ...]  -- like instep_msda.sh, but the step of scripts/ab_step.py with the given switches
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/im
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/im -- python3 $root/scripts/ab_step.py "$@" 2>&1 | grep -E "ms|images" | tail -3
python3 - <<'PY'
import csv, glob, re
for f in glob.glob("/tmp/im/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    print("  total kernel time %.1f ms" % (sum(float(r["TotalDurationNs"]) for r in rows) / 1e6))
    for r in rows:
        import os
        if re.search(os.environ.get("KPAT", r"msda_(fwd_plan|bwd_tile_accum|bwd_fold\()"), r["Name"]) or "zira::msda_bwd_fold" in r["Name"]:
            print("  %-60s calls %5s avg %8.2f us  min %7.2f max %8.2f" % (re.sub(r"\(zira_.*|\(float.*", "", r["Name"])[-60:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
