mkdir -p gpurun_out
ZIRA_SAVE_INPUTS=/tmp/dec_inputs.pt ZIRA_SAVE_ONLY=1 timeout 600 python scripts/inmodel_msda.py > gpurun_out/save.log 2>&1
for key in "$@"; do
  echo "== stamps $key"
  if [ "$key" = uniform ]; then
    ZIRA_MSDA_LIB=$PWD/build_ab/ab9.so timeout 300 python scripts/k2_stamps.py 2>&1 | grep -v amdgpu.ids
  else
    ZIRA_MSDA_LIB=$PWD/build_ab/ab9.so timeout 300 python scripts/k2_stamps.py /tmp/dec_inputs.pt $key 2>&1 | grep -v amdgpu.ids
  fi
done
