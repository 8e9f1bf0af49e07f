# usage (GPU box): scripts/kstats_case.sh <kbench case> [lib ...]  -- rocprofv3 kernel stats of one kbench case per library build
root=${GRAFT_REPO_ROOT:-$(pwd)}
c=$1; shift
cd /tmp && export TMPDIR=/tmp
for lib in "${@:-default}"; do
  if [ "$lib" = default ]; then unset ZIRA_MSDA_LIB; else export ZIRA_MSDA_LIB=$root/build_ab/$lib.so; fi
  rm -rf /tmp/ks
  CASES=$c ROUNDS=2 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 $root/scripts/kbench.py > /dev/null 2>&1
  echo "== $c $lib"
  python3 - <<'PY'
import csv, glob, re
for f in glob.glob("/tmp/ks/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "msda" in r["Name"]:
            print("  %-44s calls %5s avg %8.2f us  min %8.2f  max %8.2f" % (re.sub(r"\(.*", "", r["Name"].replace("(anonymous namespace)::", "").replace("void ", ""))[:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
