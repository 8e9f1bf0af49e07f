#!/bin/bash
# usage: [EXTRA_SRC=dev/msda_patch.hip] scripts/build_variant.sh <name> [-DZIRA_...=v ...]  -> build_ab/<name>.so (developer A/B builds; run here, they travel with gpurun)
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/build_ab
src=$root/ziragroundingdino_amd/csrc
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -munsafe-fp-atomics -I$root/include "$@" \
  $src/msda.hip $src/msda_cells.hip $src/msda_tiles.hip $src/msda_cpu.cpp $src/rsb.hip $src/xty.hip $src/bisoftmax.hip $src/layernorm.hip $src/lsap.hip $src/catlogits.hip $src/winattn.hip $src/refpoints.hip $src/attn.hip $src/sampling.hip $src/gemm_drelu.hip $src/rowgemm.hip $src/gemm_bf16x3.hip $src/gemm_f16x2.hip $src/gemm_f16x2_panel.hip $src/ffn_f16x2.hip $src/criterion.hip $src/textside.hip ${EXTRA_SRC:+$src/$EXTRA_SRC} -o $root/build_ab/$name.so
