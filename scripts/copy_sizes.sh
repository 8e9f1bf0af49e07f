# usage (GPU box): scripts/copy_sizes.sh  -- elementwise copy / add / fill kernels of the replayed bench steps grouped by grid size
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cs
rocprofv3 --kernel-trace --output-format csv -d /tmp/cs -- python3 $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-micro --no-second-mode --no-gemm-arith-mode --regions 1 > /dev/null 2>&1
python3 - > $root/gpurun_out/copy_sizes.txt <<'PY'
import csv, glob, collections, re
rows = []
for f in glob.glob("/tmp/cs/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
for pat, label in (("copyBuffer", "hipMemcpy D2D (tensor.copy_ of contiguous tensors, graph static inputs)"), ("direct_copy", "copies"), ("CUDAFunctor_add", "adds"), ("FillFunctor", "fills"), ("MulFunctor", "muls")):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        if pat in r["Kernel_Name"]:
            g = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)))
            a = agg[g]
            a[0] += 1
            a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot = sum(v[1] for v in agg.values())
    print("== %s: %.2f ms total over %d launches (whole run: 13 replayed steps + captures)" % (label, tot / 1e3, sum(v[0] for v in agg.values())))
    for g, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print("   grid %9d  x%-5d  %8.1f us total  avg %7.2f us" % (g, n, t, t / n))
PY
