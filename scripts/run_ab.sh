# usage: scripts/run_ab.sh <cases> <lib.so>...   (A/B of developer builds on the same box)
mkdir -p gpurun_out
cases=$1; shift
[ -n "$ZIRA_TESTS" ] && timeout 900 python -m pytest tests/test_msda_gpu.py -x -q -m gpu 2>&1 | tail -5
ZIRA_SAVE_INPUTS=/tmp/dec_inputs.pt ZIRA_SAVE_ONLY=1 timeout 600 python scripts/inmodel_msda.py > gpurun_out/save.log 2>&1
for lib in "$@"; do
  echo "== $lib"
  ZIRA_INPUTS=/tmp/dec_inputs.pt CASES=$cases ROUNDS=5 ZIRA_MSDA_LIB=$PWD/$lib timeout 600 python scripts/kbench.py 2>&1 | grep -v amdgpu.ids
done
