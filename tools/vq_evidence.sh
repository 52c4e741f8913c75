#!/bin/bash
# Runs ON THE GPU BOX: everything DESIGN 3.2's budget table of the VQ argmin kernels quotes, in one call, as text files under
# gpurun_out/vq_<tag>/ (copied to profiles/<tag>_vq_*.txt by hand afterwards):
#   issue_rate.txt    tools/microbench/issue_rate.hip: instructions per cycle of one SIMD by mix and waves per SIMD
#   check_time.txt    parity of kernel 17 at the probe sizes + its event-train time beside kernel 16
#   diag.txt          phase stamps (prologue / loop / last merges / refine) of every schedule variant and timing-only ablation
#   wave_stamps.txt   per-wave timeline of two periods of the tile loop
#   sq_16.txt / sq_17.txt   SQ counters of the two kernels (tools/vq_pmc.sh)
set -u
TAG=${1:-r06}
OUT=gpurun_out/vq_$TAG
mkdir -p $OUT tools/microbench/bin
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/microbench/issue_rate.hip -o tools/microbench/bin/issue_rate 2> $OUT/issue_rate.build.log
timeout 120 tools/microbench/bin/issue_rate > $OUT/issue_rate.txt 2>&1; echo "issue_rate rc=$?"
timeout 400 python3 tools/vq_pipe_check.py full adv time 2>&1 | grep -v "amdgpu.ids" > $OUT/check_time.txt; echo "check rc=$?"
timeout 900 python3 tools/vq_pipe_diag.py 2>&1 | grep -v "amdgpu.ids" > $OUT/diag.txt; echo "diag rc=$?"
timeout 200 python3 tools/vq_pipe_wave_stamps.py 1 2>&1 | grep -v "amdgpu.ids" > $OUT/wave_stamps.txt; echo "stamps rc=$?"
DVQ_VQ_KERNEL=16 bash tools/vq_pmc.sh vq_$TAG/pmc16 > $OUT/sq_16.txt 2>&1; echo "pmc16 rc=$?"
DVQ_VQ_KERNEL=17 bash tools/vq_pmc.sh vq_$TAG/pmc17 > $OUT/sq_17.txt 2>&1; echo "pmc17 rc=$?"
