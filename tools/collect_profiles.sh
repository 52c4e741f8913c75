#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats of the default bench and of the VQ microbench, plus the
# two PMC passes (FETCH_SIZE, WRITE_SIZE - separate runs, counters only with --kernel-trace) for HBM traffic.
# Outputs land in gpurun_out/prof_<tag>/; tools/summarize_profiles.py turns them into the files kept under profiles/.
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
run() { # name, then the rocprofv3 args
  local name=$1; shift
  timeout 900 rocprofv3 "$@" -d $OUT/$name -o $name --output-format csv -- python3 bench.py ${BENCH_ARGS} > $OUT/$name.json 2> $OUT/$name.log
  echo "$name rc=$?"
}
BENCH_ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-latency --prof-steps 0" run bench_stats --kernel-trace --stats
# the same with the PointNet on one stream: every kernel alone on the chip -- the durations bench.py's roofline objects quote
export DVQ_PN_STREAMS=0
BENCH_ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-latency --prof-steps 0" run bench_stats_serial --kernel-trace --stats
unset DVQ_PN_STREAMS
BENCH_ARGS="--vq-only" run vq_stats --kernel-trace --stats
BENCH_ARGS="--vq-only" run vq_pmc_fetch --kernel-trace --pmc FETCH_SIZE
BENCH_ARGS="--vq-only" run vq_pmc_write --kernel-trace --pmc WRITE_SIZE
BENCH_ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-latency --prof-steps 0" run bench_pmc_fetch --kernel-trace --pmc FETCH_SIZE
BENCH_ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-latency --prof-steps 0" run bench_pmc_write --kernel-trace --pmc WRITE_SIZE
timeout 300 python3 bench.py --vq-only --vq-tie-prone > $OUT/vq_tie_prone.json 2> $OUT/vq_tie_prone.log; echo "tie_prone rc=$?"
find $OUT -name "*.csv" | head -30
