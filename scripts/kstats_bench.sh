#!/bin/bash
# usage (on the GPU box, through gpurun): scripts/kstats_bench.sh <tag> [bench.py flags ...] -> gpurun_out/<tag>_bench_kernel_stats.txt
# rocprofv3 kernel-trace stats of a short bench.py run (step 1 of scripts/collect_profiles.sh alone, with extra flags).
root=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace_$tag -- python3 $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-micro --no-gemm-arith-mode --no-accuracy "$@" > $out/${tag}_bench_under_rocprof.log 2>&1
python3 - "$out" "$tag" "$*" <<'PY'
import csv, glob, sys, re
out, tag, flags = sys.argv[1:4]
rows = []
for f in glob.glob("/tmp/trace_%s/**/*kernel_stats.csv" % tag, recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open("%s/%s_bench_kernel_stats.txt" % (out, tag), "w") as fh:
    fh.write("rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-micro --no-gemm-arith-mode --no-accuracy %s\n" % flags)
    fh.write("total kernel time %.1f ms over %d kernel names (3 warm-up steps, 3 timed regions of 10 steps in the configured launch mode and as many in the other one, the capture of\n"
             "the graphs -- 3 warm-up passes and one capture pass per piece -- and 1 + 4 eager steps for the MSDA event timing)\n" % (tot / 1e6, len(rows)))
    fh.write("%7s %11s %8s %12s  %s\n" % ("share", "total_ms", "calls", "avg_us", "kernel"))
    for r in rows[:70]:
        name = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", r["Name"])[:110]
        fh.write("%6.2f%% %11.3f %8s %12.2f  %s\n" % (100 * float(r["TotalDurationNs"]) / tot, float(r["TotalDurationNs"]) / 1e6, r["Calls"], float(r["AverageNs"]) / 1e3, name))
    fh.write("\nhand-written kernels of this package (all of them, whatever their rank):\n")
    for r in rows:
        if re.search(r"msda_|rsb_|xty_|bis_|rowgemm|box_refine|decoder_prep|gemm_nn_drelu|gemm_bf16x3|split_bf16x3|ffn_f16x2|gemm_f16x2|split_f16x2|thin_f16x2|thin_split|sine_pos|ln_fwd_rows|ln_bwd_rows|attn_fwd|attn_bwd|window_attn|lsap|match_cost|cat_logits|sine_embed|sampling_fwd|sampling_bwd|attn_sum_parts|text_prep|text_out|text_colsum|text_ln|focal_fwd|losses_|level_counts|encoder_ref_points|encoder_proposals|box_head", r["Name"]):
            name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"]).split("(")[0]
            fh.write("%6.2f%% %11.3f %8s %12.2f  %s\n" % (100 * float(r["TotalDurationNs"]) / tot, float(r["TotalDurationNs"]) / 1e6, r["Calls"], float(r["AverageNs"]) / 1e3, name))
PY
rm -rf /tmp/trace_$tag
tail -2 $out/${tag}_bench_under_rocprof.log | cut -c1-300
