for M in 65568 65569 65600 69632 70001 131072 131073 200000; do timeout 120 python tools/vq_pipe_probe.py $M 2>&1 | grep -v amdgpu.ids | tail -2; done
echo "--- diag lib, not interleaved"
for M in 70001; do DVQ_DIAG_LIB=1 DVQ_VQP_VAR=0 timeout 120 python tools/vq_pipe_probe.py $M 2>&1 | grep -v amdgpu.ids | tail -2; done
