#!/bin/bash
# Runs ON THE GPU BOX: the step against the number of samples per PointNet launch (DVQ_PN_CHUNK; default: what 6.5 GB of scratch hold).
set -u
for c in 0 5462 4096 3641 2048; do
  DVQ_PN_CHUNK=$c timeout 600 python3 bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --no-latency 2>/dev/null | python3 -c "
import sys,json;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('chunk', $c, round(d['ms_per_step'],1), d['gathered_sha256'][:10], {n:round(k['ms'],1) for n,k in d['kernels'].items() if n.startswith('pn_') or n=='gemm_bias'})"
done
