#!/bin/bash
# timing-only ablations of the filtered PointNet trunk (diagnostics build: make -C d-vqvae_amd/csrc diag); results are INVALID.
# Read the pn_trunk column only: the exact stage evaluates garbage records exhaustively (hundreds of ms) -- one stream, so that it does
# not sit inside the next launch's trunk timing.
export DVQ_PN_STREAMS=0
for abl in ${ABLS:-0 256 1024 512 128 2 8 1 384 1408 1920}; do
  echo "== DVQ_PN_ABL=$abl"
  DVQ_DIAG_LIB=1 DVQ_PN_ABL=$abl PN_REP=2 python3 tools/pn_quick.py 2>&1 | grep "^C="
done
