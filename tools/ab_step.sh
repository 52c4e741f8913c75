#!/bin/bash
# Runs ON THE GPU BOX: the step of this tree against the tree in tools/ab/base (git worktree, built there), alternating, same box.
# Prints ms per step, the big kernels and the result hash of each run.
set -u
OUT=gpurun_out/${1:-ab_step}
mkdir -p $OUT
for rep in 1 2; do
  for v in new base; do
    if [ $v = base ]; then b=tools/ab/base/bench.py; else b=bench.py; fi
    timeout 600 python3 $b --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --no-latency > $OUT/$v$rep.json 2> $OUT/$v$rep.err
    python3 - $OUT/$v$rep.json $v$rep <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ks = d.get("kernels", {})
print(sys.argv[2], "ms/step", round(d["ms_per_step"], 1), "sha", d.get("gathered_sha256", "")[:12], {n: round(k["ms"], 1) for n, k in ks.items() if k["ms"] > 5})
PY
  done
done
