#!/bin/bash
export TMPDIR=/tmp
rm -rf /tmp/pnprof
PN_B=8192 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pnprof -o pn --output-format csv -- python3 tools/pn_filter_bench.py > /tmp/pnprof.log 2>&1
grep -E "encode|dvq pn" /tmp/pnprof.log
f=$(find /tmp/pnprof -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print(r["Name"][:70], r["Calls"], "avg us %.1f" % (float(r["AverageNs"]) / 1e3), "total ms %.1f" % (float(r["TotalDurationNs"]) / 1e6))
PY
