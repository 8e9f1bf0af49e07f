# usage: scripts/ab_libs.sh <bwd path> <lib name|default> ...  -- times kbench cases (CASES env, default: decoder + in-model decoder layers) per library build
mkdir -p gpurun_out
[ -f /tmp/inmodel.pt ] || ZIRA_SAVE_ALL_DEC=1 ZIRA_SAVE_INPUTS=/tmp/inmodel.pt ZIRA_SAVE_ONLY=1 timeout 900 python scripts/inmodel_msda.py > gpurun_out/save.log 2>&1 || tail -5 gpurun_out/save.log
path=$1; shift
for lib in "$@"; do
  echo "== lib=$lib ZIRA_MSDA_BWD=$path"
  if [ "$lib" = default ]; then unset ZIRA_MSDA_LIB; else export ZIRA_MSDA_LIB=$PWD/build_ab/$lib.so; fi
  ZIRA_MSDA_BWD=$path ZIRA_INPUTS=/tmp/inmodel.pt CASES=${CASES:-decoder,inmodel_dec} ROUNDS=5 timeout 600 python scripts/kbench.py 2>&1 | grep -E "decoder|inmodel"
done
