#!/bin/bash
# usage (GPU box): scripts/pmc_py.sh <tag> <kernel-name regex> <script.py> [args]
# PMC counters (separate passes, never combined with tracing) of the kernels whose name matches, summed per kernel name,
# into gpurun_out/pmc_<tag>.txt; PMC_EXTRA1 / PMC_EXTRA2 = further counter sets (one pass each), e.g. "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
tag=$1; pat=$2; shift; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
script=$1; shift
case "$script" in /*) ;; *) script=$root/$script ;; esac
set -- "$script" "$@"
out=/tmp/pmc_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" ${PMC_EXTRA1:+"$PMC_EXTRA1"} ${PMC_EXTRA2:+"$PMC_EXTRA2"}; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $out/p$i -- python3 "$@" > $out/p$i.log 2>&1 || tail -3 $out/p$i.log
done
python3 - "$out" "$pat" > $root/gpurun_out/pmc_$tag.txt <<'PY'
import csv, glob, sys, collections, re
out, pat = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if re.search(pat, r["Kernel_Name"]):
            k = re.sub(r"^void |\(anonymous namespace\)::", "", r["Kernel_Name"])
            k = re.sub(r"\(.*", "", k)[:90]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("  %-34s n=%4d mean=%16.1f" % (c, len(v), sum(v) / len(v)))
PY
cat $root/gpurun_out/pmc_$tag.txt
