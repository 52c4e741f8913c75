#!/bin/bash
export TMPDIR=/tmp
# needs the diagnostics build: make -C d-vqvae_amd/csrc diag
for abl in ${ABLS:-0}; do
  rm -rf /tmp/pnprof
  DVQ_DIAG_LIB=1 DVQ_PN_ABL=$abl PN_B=4096 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pnprof -o pn --output-format csv -- python3 tools/pn_filter_bench.py > /tmp/pnprof.log 2>&1
  f=$(find /tmp/pnprof -name "*kernel_stats.csv" | head -1)
  python3 - "$f" $abl <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
out = []
for r in rows:
    for k in ("pn_trunk_filter", "pn_exact", "pn_trunk_kernel", "pn_center"):
        if k in r["Name"]: out.append("%s %.0f us" % (k, float(r["AverageNs"]) / 1e3))
print("abl", sys.argv[2], "|", " | ".join(out))
PY
  grep -E "encode|diff" /tmp/pnprof.log
done
