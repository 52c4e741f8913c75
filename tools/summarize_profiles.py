#!/usr/bin/env python3
"""Turn the rocprofv3 output of tools/collect_profiles.sh (gpurun_out/prof_<tag>/) into the summaries kept under
profiles/: the per-kernel stats CSVs as they are, and <tag>_pmc_hbm_traffic.json = HBM bytes per launch per kernel from the
FETCH_SIZE and WRITE_SIZE passes (both count KB; FETCH_SIZE x2 on gfx950, /opt/skills/guides/MI355X_MICROARCH.md).

    python tools/summarize_profiles.py r01_v4
"""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find(d, pattern):
    hits = sorted(glob.glob(os.path.join(d, "**", pattern), recursive=True))
    return hits[0] if hits else None


def counters(path, name):
    acc = defaultdict(lambda: [0.0, 0])
    if not path:
        return acc
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != name:
                continue
            a = acc[row["Kernel_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
    return acc


def traffic(src, stem):
    fe = counters(find(os.path.join(src, f"{stem}_pmc_fetch"), "*counter_collection.csv"), "FETCH_SIZE")
    wr = counters(find(os.path.join(src, f"{stem}_pmc_write"), "*counter_collection.csv"), "WRITE_SIZE")
    rows = []
    for k in sorted(fe, key=lambda k: -fe[k][0]):
        n = fe[k][1]
        fb = fe[k][0] / n * 1024.0
        wb = (wr[k][0] / wr[k][1] * 1024.0) if k in wr and wr[k][1] else 0.0
        rows.append({"kernel": k[:60], "launches": n, "fetch_size_bytes_raw": fb, "write_size_bytes": wb,
                     "hbm_bytes_corrected": 2.0 * fb + wb})
    return rows


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    for stem in ("bench_stats", "bench_stats_serial", "vq_stats"):
        p = find(os.path.join(src, stem), "*kernel_stats.csv")
        if p:
            shutil.copy(p, os.path.join(dst, f"{tag}_{stem}_rocprofv3_kernel_stats.csv"))
        j = os.path.join(src, f"{stem}.json")
        if os.path.exists(j) and os.path.getsize(j):
            shutil.copy(j, os.path.join(dst, f"{tag}_{stem}_bench_under_rocprofv3.json"))
    tp = os.path.join(src, "vq_tie_prone.json")
    if os.path.exists(tp) and os.path.getsize(tp):
        shutil.copy(tp, os.path.join(dst, f"{tag}_vq_tie_prone_regime.json"))
    a, b = find(os.path.join(src, "bench_stats"), "*kernel_stats.csv"), find(os.path.join(src, "bench_stats_7steps"), "*kernel_stats.csv")
    if a and b:
        # launches PER STEP of every kernel = (calls in the 7-step run - calls in the 3-step run) / 4: start-up work (weight uploads, packing,
        # the VQ microbench) cancels.  The runtime's own copy / fill kernels are listed first.
        ca = {r["Name"]: int(r["Calls"]) for r in csv.DictReader(open(a))}
        cb = {r["Name"]: int(r["Calls"]) for r in csv.DictReader(open(b))}
        per = {k: (cb.get(k, 0) - ca.get(k, 0)) / 4.0 for k in sorted(set(ca) | set(cb))}
        rt = {k: {"calls_3_steps": ca.get(k, 0), "calls_7_steps": cb.get(k, 0), "per_step": v} for k, v in per.items() if "rocclr" in k}
        json.dump({"how": "rocprofv3 --kernel-trace --stats of bench.py with --warmup 1 --steps 2 and --warmup 1 --steps 6; per_step = difference / 4",
                   "runtime_copy_fill_kernels": rt, "runtime_copy_fill_launches_per_step": sum(v["per_step"] for v in rt.values()),
                   "all_launches_per_step": sum(per.values()),
                   "launches_per_step_by_kernel": {k[:100]: v for k, v in sorted(per.items(), key=lambda kv: -kv[1]) if v}},
                  open(os.path.join(dst, f"{tag}_launches_per_step.json"), "w"), indent=1)
    for stem in ("config3", "config4", "bench"):
        j = os.path.join(src, f"{stem}.json")
        if os.path.exists(j) and os.path.getsize(j):
            shutil.copy(j, os.path.join(dst, f"{tag}_bench_{stem}.json" if stem != "bench" else f"{tag}_bench.json"))
    rows = traffic(src, "bench") + [dict(r, scope="vq microbench") for r in traffic(src, "vq")]
    if rows:
        import subprocess, datetime
        try:
            head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
        except Exception:
            head = "?"
        dg = os.path.join(src, "kernel_sources.sha256")
        digest = open(dg).read().strip() if os.path.exists(dg) else None
        json.dump({"collected": f"tag {tag}, tree at/after commit {head}, {datetime.date.today().isoformat()}", "commit": head,
                   "kernel_sources_sha256": digest, "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only), KB counters "
                           "converted to bytes, mean per launch; per MI355X_MICROARCH.md FETCH_SIZE under-reports wide "
                           "coalesced reads by 2x on gfx950 (x2 applied in hbm_bytes_corrected)",
                   "per_launch_bytes": rows}, open(os.path.join(dst, f"{tag}_pmc_hbm_traffic.json"), "w"), indent=1)
    print("wrote", sorted(f for f in os.listdir(dst) if f.startswith(tag)))


if __name__ == "__main__":
    main()
