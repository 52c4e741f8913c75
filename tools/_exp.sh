# correctness of every structure variant (diagnostics library), then timings
for v in 1 2 3 4 5 7; do DVQ_DIAG_LIB=1 DVQ_VQP_VAR=$v timeout 120 python tools/vq_pipe_probe.py 70001 2>&1 | grep -v "amdgpu.ids\|DIAGNOSTICS" | tail -1; done
VQP_STAGES="small full" VQP_STAMP_VARS="0 1 3" VQP_DIAG_ONLY=var bash tools/vq_pipe_gpu.sh
