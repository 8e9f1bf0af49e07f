#!/usr/bin/env python3
"""Dev: kernel time of the bench command by family, from a profiles/*_bench_kernel_stats.txt-style csv directory or the
summary text: `python scripts/step_categories.py gpurun_out/profiles_r05/bench_kernel_stats.txt`."""
import re
import sys

rows = []
for line in open(sys.argv[1]):
    m = re.match(r"\s*([\d.]+)%\s+([\d.]+)\s+(\d+)\s+([\d.]+)\s+(.*)", line)
    if m:
        rows.append((float(m.group(2)), int(m.group(3)), m.group(5)))
    if line.startswith("hand-written kernels"):
        break
fam = {}
def family(n):
    if n.startswith("Cijk_"): return "library GEMM"
    if "gemm_bf16x3" in n or "gemm_nn_drelu" in n or "rowgemm" in n or "gemm_f16x2" in n or "ffn_f16x2" in n: return "own GEMM"
    if "msda_" in n: return "MSDA"
    if re.search(r"direct_copy|copyBuffer|CatArray|fillBuffer|FillFunctor", n): return "ATen copy / fill / cat"
    if re.search(r"elementwise|vectorized_|reduce_kernel|layer_norm|softmax|GroupNorm|group_norm|index|gather|scatter|sort|topk|multi_tensor", n): return "ATen elementwise / norm / reduce"
    return "own other"
tot = sum(r[0] for r in rows)
for t, c, n in rows:
    f = fam.setdefault(family(n), [0.0, 0])
    f[0] += t
    f[1] += c
print("listed kernel time %.1f ms" % tot)
for k, (t, c) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
    print("%-36s %8.1f ms %5.1f %%  %7d launches" % (k, t, 100 * t / tot, c))
