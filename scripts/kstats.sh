# usage (GPU box): scripts/kstats.sh <cases> [lib]  -- rocprofv3 kernel stats of kbench on the given cases
mkdir -p gpurun_out
root=${GRAFT_REPO_ROOT:-$(pwd)}
[ -f /tmp/inmodel.pt ] || ZIRA_SAVE_ALL_DEC= ZIRA_SAVE_INPUTS=/tmp/inmodel.pt ZIRA_SAVE_ONLY=1 timeout 900 python scripts/inmodel_msda.py > gpurun_out/save.log 2>&1
[ -n "$2" ] && export ZIRA_MSDA_LIB=$root/build_ab/$2.so
cd /tmp && export TMPDIR=/tmp
for c in ${1//,/ }; do
  rm -rf /tmp/ks
  ZIRA_INPUTS=/tmp/inmodel.pt CASES=$c ROUNDS=2 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 $root/scripts/kbench.py > /dev/null 2>&1
  echo "== $c ${2:-default}"
  python3 - <<'PY'
import csv, glob, re
for f in glob.glob("/tmp/ks/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "msda" in r["Name"]:
            print("  %-44s calls %5s avg %8.2f us  min %8.2f  max %8.2f" % (re.sub(r"\(.*", "", r["Name"].replace("(anonymous namespace)::", "").replace("void ", ""))[:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
