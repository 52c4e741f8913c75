#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of a microbench binary's kernels (wave cycles split into waiting / issue-stalled / active, LDS
# conflicts, MFMA busy, clock), passes of <= 8 SQ counters, counters only with --kernel-trace.  usage: tools/mb_pmc.sh <tag> <binary> [args...]
set -u
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/a -o a --output-format csv -- "$@" > $OUT/a.out 2> $OUT/a.log
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU -d $OUT/b -o b --output-format csv -- "$@" > $OUT/b.out 2> $OUT/b.log
python3 - <<'PY' $OUT
import csv, glob, sys, collections
out = sys.argv[1]
dur = collections.defaultdict(lambda: [0.0, 0])
for p in glob.glob(f"{out}/a/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(p)):
        d = dur[row["Kernel_Name"][:60]]; d[0] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"]); d[1] += 1
for sub in ("a", "b"):
    for p in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(p)):
            key = (row["Kernel_Name"][:60], row["Counter_Name"])
            a = acc[key]; a[0] += float(row["Counter_Value"]); a[1] += 1
        for k, (v, n) in sorted(acc.items()):
            print(f"{k[0]:60s} {k[1]:30s} per launch {v / n:16.0f}   ({n} launches)")
for k, (t, n) in sorted(dur.items()):
    print(f"{k:60s} avg duration {t / n * 1e-3:10.1f} us ({n} launches, profiled pass a)")
PY
