#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of the PointNet trunk kernels (pn_trunk_filter_kernel, pn_exact_kernel, pn_center_kernel)
# under tools/pn_filter_bench.py: wave cycles split into waiting / issue-stalled / active, instruction counts by kind, matrix
# pipe busy cycles, LDS conflicts.  Passes of <= 8 SQ counters, counters only with --kernel-trace (no other trace domain).
# Usage: bash tools/pn_pmc.sh <out-subdir> ; the JSON summary it prints is what is kept under profiles/.
set -u
OUT=gpurun_out/${1:-pn_pmc}
mkdir -p $OUT
export TMPDIR=/tmp
export PN_B=${PN_B:-4096}
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $OUT/a -o a --output-format csv -- python3 tools/pn_filter_bench.py > $OUT/a.txt 2> $OUT/a.log
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU -d $OUT/b -o b --output-format csv -- python3 tools/pn_filter_bench.py > $OUT/b.txt 2> $OUT/b.log
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM GRBM_GUI_ACTIVE -d $OUT/c -o c --output-format csv -- python3 tools/pn_filter_bench.py > $OUT/c.txt 2> $OUT/c.log
python3 - <<'PY' $OUT
import csv, glob, sys, collections, json
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
dur = collections.defaultdict(lambda: [0.0, 0])
for sub in ("a", "b", "c"):
    for p in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(p)):
            name = row["Kernel_Name"]
            if "pn_" not in name: continue
            key = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            a = acc[key][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
    for p in glob.glob(f"{out}/{sub}/**/*kernel_trace.csv", recursive=True):
        if sub != "a": continue
        for row in csv.DictReader(open(p)):
            name = row["Kernel_Name"]
            if "pn_" not in name: continue
            key = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            d = dur[key]; d[0] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3; d[1] += 1
doc = {"command": f"rocprofv3 --kernel-trace --pmc <8 SQ counters> -- python3 tools/pn_filter_bench.py (PN_B={__import__('os').environ.get('PN_B')}, N=1024, C=4), three passes",
       "units": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES count cycles (per SE / XCD sums as rocprofv3 reports them); per launch averages",
       "kernels": {}}
for k, cs in sorted(acc.items()):
    e = {c: v / n for c, (v, n) in sorted(cs.items())}
    e["launches_per_pass"] = max(n for _, n in cs.values())
    if k in dur: e["avg_duration_us_under_pmc"] = dur[k][0] / dur[k][1]
    wc = e.get("SQ_WAVE_CYCLES")
    if wc:
        e["share_of_wave_cycles"] = {"waiting (s_waitcnt / barrier)": e.get("SQ_WAIT_ANY", 0) / wc, "issue-stalled": e.get("SQ_WAIT_INST_ANY", 0) / wc,
                                     "issuing": e.get("SQ_ACTIVE_INST_ANY", 0) / wc, "issuing VALU": e.get("SQ_ACTIVE_INST_VALU", 0) / wc,
                                     "issuing LDS": e.get("SQ_ACTIVE_INST_LDS", 0) / wc}
    if e.get("SQ_BUSY_CYCLES") and e.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        e["mfma_busy_over_sq_busy"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / e["SQ_BUSY_CYCLES"]
    doc["kernels"][k] = e
print(json.dumps(doc, indent=1))
PY
tail -2 $OUT/a.log $OUT/b.log $OUT/c.log | cut -c1-200 >&2
