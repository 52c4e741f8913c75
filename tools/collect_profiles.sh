#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats of the default bench and of the VQ microbench, plus the
# two PMC passes (FETCH_SIZE, WRITE_SIZE - separate runs, counters only with --kernel-trace) for HBM traffic.
# Outputs land in gpurun_out/prof_<tag>/; tools/summarize_profiles.py turns them into the files kept under profiles/.
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py --sources-digest > $OUT/kernel_sources.sha256      # bench.py quotes these counters only on the same kernel sources
run() { # name, then the rocprofv3 args
  local name=$1; shift
  timeout 900 rocprofv3 "$@" -d $OUT/$name -o $name --output-format csv -- python3 bench.py ${BENCH_ARGS} > $OUT/$name.json 2> $OUT/$name.log
  echo "$name rc=$?"
}
BENCH_ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-latency --prof-steps 0" run bench_stats --kernel-trace --stats
# the same command with four more steps: what the runtime's copy / fill kernels (__amd_rocclr_copyBuffer: the host-to-device uploads of
# the weights at start-up use it too) cost PER STEP is the difference of the two runs' call counts / 4
BENCH_ARGS="--steps 6 --warmup 1 --no-cpu-baseline --no-latency --prof-steps 0" run bench_stats_7steps --kernel-trace --stats
# the same with the PointNet on one stream: every kernel alone on the chip -- the durations bench.py's roofline objects quote
export DVQ_PN_STREAMS=0
BENCH_ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-latency --prof-steps 0" run bench_stats_serial --kernel-trace --stats
unset DVQ_PN_STREAMS
BENCH_ARGS="--vq-only" run vq_stats --kernel-trace --stats
BENCH_ARGS="--vq-only" run vq_pmc_fetch --kernel-trace --pmc FETCH_SIZE
BENCH_ARGS="--vq-only" run vq_pmc_write --kernel-trace --pmc WRITE_SIZE
BENCH_ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-latency --prof-steps 0" run bench_pmc_fetch --kernel-trace --pmc FETCH_SIZE
BENCH_ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-latency --prof-steps 0" run bench_pmc_write --kernel-trace --pmc WRITE_SIZE
# BASELINE configs 3 and 4 (single-GPU legs: a ragged 3000-point cloud, and the seeded B = 1 loop) and the driver's own command
timeout 600 python3 bench.py --config 3 > $OUT/config3.json 2> $OUT/config3.log; echo "config3 rc=$?"
timeout 600 python3 bench.py --config 4 > $OUT/config4.json 2> $OUT/config4.log; echo "config4 rc=$?"
if [ -z "${NO_DEFAULT_BENCH:-}" ]; then timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.log; echo "bench rc=$?"; fi
timeout 300 python3 bench.py --vq-only --vq-tie-prone > $OUT/vq_tie_prone.json 2> $OUT/vq_tie_prone.log; echo "tie_prone rc=$?"
find $OUT -name "*.csv" | head -30
