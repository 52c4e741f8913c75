#!/bin/bash
# Runs ON THE GPU BOX: the step's time against the PixelCNN chunk size (rows of the batch walked through the 3x3 grid together).
set -u
OUT=gpurun_out/${1:-chunk_sweep}
mkdir -p $OUT
for c in 8192 16384 32768 65536; do
  DVQ_PIXELCNN_CHUNK=$c timeout 600 python3 bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --no-latency > $OUT/c$c.json 2> $OUT/c$c.err
  python3 - $OUT/c$c.json $c <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ks = d.get("kernels", {})
print("chunk", sys.argv[2], "ms/step", round(d["ms_per_step"], 1), "sha", d.get("gathered_sha256", "")[:12],
      {n: round(k["ms"], 1) for n, k in ks.items() if n.startswith("gemm_") and k["ms"] > 5})
PY
done
