#!/bin/bash
# same-box A/B of the PointNet kernels: this tree against tools/ab/base (git worktree add -f tools/ab/base <commit> && make -C
# tools/ab/base/d-vqvae_amd/csrc -j8; tools/ab/ is git-ignored), alternating, every kernel alone on the chip (one stream).
export DVQ_PN_STREAMS=0
for rep in 1 2; do
  for v in new base; do
    if [ $v = base ]; then q=tools/ab/base/tools/pn_quick.py; else q=tools/pn_quick.py; fi
    echo "== $v$rep"; python3 $q 2>/dev/null
  done
done
