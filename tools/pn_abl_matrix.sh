#!/bin/bash
# timing-only ablations of the filtered PointNet trunk (diagnostics build: make -C d-vqvae_amd/csrc diag); results are INVALID
for abl in ${ABLS:-0 256 1024 512 128 2 8 1 384 1408 1920}; do
  echo "== DVQ_PN_ABL=$abl"
  DVQ_DIAG_LIB=1 DVQ_PN_ABL=$abl PN_REP=2 python3 tools/pn_quick.py 2>&1 | grep "^C="
done
