#!/bin/bash
# Runs ON THE GPU BOX: vq_pipe.hip stage by stage, each under its own timeout; then the diagnostics
mkdir -p gpurun_out/vqp
for st in ${VQP_STAGES:-small mid full adv time}; do
  echo "=== stage $st"; timeout 240 python tools/vq_pipe_check.py $st 2>&1 | grep -v amdgpu.ids | tail -25
  rc=${PIPESTATUS[0]}; echo "stage $st rc=$rc"; if [ $rc -ne 0 ]; then exit 1; fi
done 2>&1 | tee gpurun_out/vqp/check.log
for v in ${VQP_STAMP_VARS:-0}; do timeout 300 python tools/vq_pipe_wave_stamps.py $v 2>&1 | grep -v "amdgpu.ids\|DIAGNOSTICS" | tee gpurun_out/vqp/stamps_$v.log | tail -22; done
timeout 900 python tools/vq_pipe_diag.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/vqp/diag.log | tail -32
