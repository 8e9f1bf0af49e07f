# usage: scripts/run_kt.sh "<prof_target args>" <lib.so>...     (ZIRA_INMODEL=1: capture model inputs first)
args=$1; shift
if [ -n "$ZIRA_INMODEL" ]; then
  ZIRA_SAVE_INPUTS=/tmp/dec_inputs.pt ZIRA_SAVE_ONLY=1 timeout 600 python scripts/inmodel_msda.py > gpurun_out/save.log 2>&1
  export ZIRA_INPUTS=/tmp/dec_inputs.pt
fi
for lib in "$@"; do echo "== $lib"; ZIRA_MSDA_LIB=$PWD/$lib bash scripts/ktrace.sh $(basename $lib .so) $args | grep -v copyBuffer; done
