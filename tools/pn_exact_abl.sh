#!/bin/bash
# timing-only ablations of pn_exact_kernel (diagnostics build): 64 = no candidate dots (phase B), 32 = no flagged groups (phase C), 16 = every candidate is one of 64 rows (rows cached)
export DVQ_PN_STREAMS=0 DVQ_DIAG_LIB=1 PN_REP=3
for a in 0 64 32 96 16; do
  echo "== DVQ_PN_ABL=$a"; DVQ_PN_ABL=$a python3 tools/pn_quick.py 2>/dev/null | sed -e 's/gemm_bias.*//'
done
