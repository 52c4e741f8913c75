#!/bin/bash
# Runs ON THE GPU BOX: first contact of vq_pipe.hip, stage by stage, each under its own timeout
mkdir -p gpurun_out/vqp
for st in small mid full adv time; do
  echo "=== stage $st"; timeout 240 python tools/vq_pipe_check.py $st 2>&1 | tail -25
  rc=${PIPESTATUS[0]}; echo "stage $st rc=$rc"; if [ $rc -ne 0 ]; then break; fi
done 2>&1 | tee gpurun_out/vqp/check.log
