#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of the VQ streaming kernel (wave cycles split into waiting / issue-stalled / active,
# instruction counts by kind), two passes of <= 8 SQ counters, counters only with --kernel-trace.
set -u
OUT=gpurun_out/${1:-vq_pmc}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $OUT/a -o a --output-format csv -- python3 bench.py --vq-only > $OUT/a.json 2> $OUT/a.log
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU -d $OUT/b -o b --output-format csv -- python3 bench.py --vq-only > $OUT/b.json 2> $OUT/b.log
python3 - <<'PY' $OUT
import csv, glob, sys, collections
out = sys.argv[1]
for sub in ("a", "b"):
    for p in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(p)):
            if "vq_stream" not in row["Kernel_Name"] and "vq_pipe" not in row["Kernel_Name"]: continue
            a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
        for k, (v, n) in sorted(acc.items()):
            print(f"{k:34s} per launch {v / n:16.0f}   ({n} launches)")
PY
tail -n 3 $OUT/a.log; tail -n 3 $OUT/b.log
