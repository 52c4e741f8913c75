#!/bin/bash
# same-box A/B: this tree's step against the round-4 tree, alternating.  Needs (here, before gpurun): git worktree add -f tools/ab/r04 d0b8d5b && make -C tools/ab/r04/d-vqvae_amd/csrc -j8   (tools/ab/ is git-ignored)
for rep in 1 2; do
  for v in new old; do
    if [ $v = old ]; then b=tools/ab/r04/bench.py; else b=bench.py; fi
    timeout 900 python3 $b --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --no-latency --no-prof 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v$rep', round(d['ms_per_step'],1), 'ms', round(d['value']), d['gathered_sha256'][:12])"
  done
done
DVQ_PN_STREAMS=0 python3 bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --no-latency --no-prof 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new, one stream', round(d['ms_per_step'],1), 'ms', d['gathered_sha256'][:12])"
