#!/bin/bash
# usage: scripts/pmc.sh <tag> <prof_target args...>   (run on the GPU box, from the repo root)
# Collects PMC counters in separate passes (never combined with tracing) into gpurun_out/pmc_<tag>/.
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum TCC_WRITE_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --output-format csv -d $out/p$i -- python3 $root/scripts/prof_target.py "$@" > $out/p$i.log 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        import re
        mm = re.search(r"(msda_\w+(<\d+>)?|__amd_\w+)", r["Kernel_Name"])
        k = mm.group(1) if mm else r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fh:
    for k, cs in agg.items():
        fh.write(k + "\n")
        for c, v in sorted(cs.items()):
            fh.write("  %-34s n=%3d mean=%16.1f\n" % (c, len(v), sum(v) / len(v)))
print(open(out + "/summary.txt").read())
PY
