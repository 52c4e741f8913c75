mkdir -p gpurun_out/vqp
timeout 300 python tools/vq_pipe_check.py full adv time 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/vqp/check.log | tail -14
VQP_DIAG_ONLY=var timeout 600 python tools/vq_pipe_diag.py 2>&1 | grep -v "amdgpu.ids\|DIAGNOSTICS" | tee gpurun_out/vqp/diag.log | tail -18
