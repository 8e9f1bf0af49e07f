# usage (GPU box): scripts/instep_msda.sh [lib ...]  -- rocprofv3 durations of the MSDA kernels INSIDE replayed training steps (bench.py), per library build
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in "${@:-default}"; do
  if [ "$lib" = default ]; then unset ZIRA_MSDA_LIB; else export ZIRA_MSDA_LIB=$root/build_ab/$lib.so; fi
  rm -rf /tmp/im
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/im -- python3 $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-micro --no-second-mode > /dev/null 2>&1
  echo "== $lib"
  python3 - <<'PY'
import csv, glob, re
for f in glob.glob("/tmp/im/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    print("  total kernel time %.1f ms" % (sum(float(r["TotalDurationNs"]) for r in rows) / 1e6))
    for r in rows:
        if re.search(r"msda_(fwd_plan|bwd_tile_accum|bwd_fold\()", r["Name"]) or "zira::msda_bwd_fold" in r["Name"]:
            print("  %-40s calls %5s avg %8.2f us" % (re.sub(r"\(.*", "", r["Name"])[:40], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
