# usage: scripts/inmodel_ab.sh  -- times the backward paths on MSDA inputs captured from a real model step
mkdir -p gpurun_out
ZIRA_SAVE_ALL_DEC=${ZIRA_SAVE_ALL_DEC:-} ZIRA_SAVE_INPUTS=/tmp/inmodel.pt ZIRA_SAVE_ONLY=1 timeout 900 python scripts/inmodel_msda.py > gpurun_out/save.log 2>&1 || tail -5 gpurun_out/save.log
for path in "$@"; do
  echo "== ZIRA_MSDA_BWD=$path ${ZIRA_MSDA_LIB:+lib=$ZIRA_MSDA_LIB}"
  ZIRA_MSDA_BWD=$path ZIRA_INPUTS=/tmp/inmodel.pt CASES=inmodel ROUNDS=5 timeout 600 python scripts/kbench.py 2>&1 | grep inmodel
done
