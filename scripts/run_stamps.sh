mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_msda_gpu.py -x -q -m gpu 2>&1 | tail -5
ZIRA_SAVE_INPUTS=/tmp/dec_inputs.pt ZIRA_SAVE_ONLY=1 timeout 600 python scripts/inmodel_msda.py > gpurun_out/save.log 2>&1
echo "== uniform stamps" > gpurun_out/stamps.log
ZIRA_MSDA_LIB=$PWD/build_ab/ab9.so timeout 300 python scripts/k2_stamps.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/stamps.log
echo "== in-model dec stamps" >> gpurun_out/stamps.log
ZIRA_MSDA_LIB=$PWD/build_ab/ab9.so timeout 300 python scripts/k2_stamps.py /tmp/dec_inputs.pt dec 2>&1 | grep -v amdgpu.ids >> gpurun_out/stamps.log
cat gpurun_out/stamps.log
timeout 600 python scripts/kbench.py 2>&1 | grep -v amdgpu.ids | tail -12
