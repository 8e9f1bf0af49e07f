#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root: collects the rocprofv3 evidence that is
# committed under profiles/ -- kernel-trace stats of the default bench.py command, and PMC
# counters (separate passes, never combined with tracing) of the MSDA kernels at the north-star
# decoder shape and at the encoder shape.
root=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r01}
out=$root/gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
# 1. kernel trace + stats of the bench command (fewer steps: the trace inflates host time)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_trace -- python3 $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-micro --no-gemm-arith-mode --no-accuracy > $out/bench_under_rocprof.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, re
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/bench_trace/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(out + "/bench_kernel_stats.txt", "w") as fh:
    fh.write("rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-micro --no-gemm-arith-mode --no-accuracy\n")
    fh.write("total kernel time %.1f ms over %d kernel names (3 warm-up steps, 3 timed regions of 10 steps in the configured launch mode and as many in the other one, the capture of\n"
             "the graphs -- 3 warm-up passes and one capture pass per piece -- and 1 + 4 eager steps for the MSDA event timing)\n" % (tot / 1e6, len(rows)))
    fh.write("%7s %11s %8s %12s  %s\n" % ("share", "total_ms", "calls", "avg_us", "kernel"))
    for r in rows[:60]:
        name = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", r["Name"])[:110]
        fh.write("%6.2f%% %11.3f %8s %12.2f  %s\n" % (100 * float(r["TotalDurationNs"]) / tot, float(r["TotalDurationNs"]) / 1e6, r["Calls"], float(r["AverageNs"]) / 1e3, name))
    fh.write("\nhand-written kernels of this package (all of them, whatever their rank):\n")
    for r in rows:
        if re.search(r"msda_|rsb_|xty_|bis_|rowgemm|box_refine|decoder_prep|gemm_nn_drelu|gemm_bf16x3|split_bf16x3|gemm_f16x2|split_f16x2|thin_f16x2|thin_split|ffn_f16x2|sine_pos|ln_fwd_rows|ln_bwd_rows|attn_fwd|attn_bwd|window_attn|lsap|match_cost|cat_logits|sine_embed|sampling_fwd|sampling_bwd|attn_sum_parts|text_prep|text_out|text_colsum|text_ln|focal_fwd|losses_|level_counts|encoder_ref_points|encoder_proposals|box_head", r["Name"]):
            name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"]).split("(")[0]
            fh.write("%6.2f%% %11.3f %8s %12.2f  %s\n" % (100 * float(r["TotalDurationNs"]) / tot, float(r["TotalDurationNs"]) / 1e6, r["Calls"], float(r["AverageNs"]) / 1e3, name))
print(open(out + "/bench_kernel_stats.txt").read()[:3000])
PY
# 2. MSDA kernels alone: kernel trace (durations) + PMC passes
for shape in decoder encoder; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/msda_${shape}_trace -- python3 $root/scripts/prof_target.py both $shape 20 > /dev/null 2>&1
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
             "TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    timeout 120 rocprofv3 --pmc $set --output-format csv -d $out/msda_${shape}_pmc/p$i -- python3 $root/scripts/prof_target.py both $shape 4 > /dev/null 2>&1
  done
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections, re, json
out = sys.argv[1]
summary = {}
for shape in ("decoder", "encoder"):
    dur = collections.defaultdict(list)
    for f in glob.glob(out + "/msda_%s_trace/**/*kernel_trace.csv" % shape, recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(msda_\w+(<[^>]*>)?|__amd_rocclr_\w+)", r["Kernel_Name"])
            if m:
                dur[m.group(1)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(out + "/msda_%s_pmc/p*/**/*counter_collection.csv" % shape, recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(msda_\w+(<[^>]*>)?)", r["Kernel_Name"])
            if m:
                agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    s = {}
    for k in agg:
        c = {n: sum(v) / len(v) for n, v in agg[k].items()}
        d = sorted(dur.get(k, [0.0]))
        # gfx950: FETCH_SIZE (KiB) counts 128-B requests at 64 B for 16-B-per-lane loads -> x2;
        # WRITE_SIZE (KiB) is exact for 16-B-per-lane stores (MI355X_MICROARCH.md, HBM section)
        c["hbm_read_bytes_corrected"] = 2 * 1024 * c.get("FETCH_SIZE", 0.0)
        c["hbm_write_bytes"] = 1024 * c.get("WRITE_SIZE", 0.0)
        c["median_us"] = d[len(d) // 2]
        s[k] = c
    summary[shape] = s
json.dump(summary, open(out + "/msda_pmc_summary.json", "w"), indent=1, sort_keys=True)
for shape, s in summary.items():
    for k, c in s.items():
        print("%-8s %-24s median %8.1f us  HBM read %.1f MB (corrected)  write %.1f MB  L2 hit %.0f%%  VALU/wave %.0f" % (
            shape, k, c["median_us"], c["hbm_read_bytes_corrected"] / 1e6, c["hbm_write_bytes"] / 1e6,
            100 * c.get("TCC_HIT_sum", 0) / max(1.0, c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)),
            c.get("SQ_INSTS_VALU", 0) / max(1.0, c.get("SQ_WAVES", 1))))
PY
rm -rf $out/bench_trace $out/msda_*_trace $out/msda_*_pmc
