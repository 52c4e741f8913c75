#!/usr/bin/env python3
"""Static instruction counts of one kernel in a `hipcc -S --cuda-device-only` listing, by class, whole kernel and per region
(regions are split at s_barrier).  usage: isa_count.py listing.s name-substring [...]"""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_mfma"): return "MFMA"
    if op.startswith("v_"): return "VALU"
    if op.startswith(("s_waitcnt", "s_nop")): return "WAIT"
    if op == "s_barrier": return "BARRIER"
    if op.startswith("s_"): return "SALU"
    if op.startswith("ds_"): return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "VMEM"
    return "OTHER"


def main():
    text = open(sys.argv[1]).read()
    for want in sys.argv[2:]:
        for m in re.finditer(r"^(\S+):\s*; @\1\n", text, re.M):
            name = m.group(1)
            if want not in name:
                continue
            end = text.index("s_endpgm", m.end())
            body = text[m.end():end].splitlines()
            total, regions, cur = Counter(), [], Counter()
            for line in body:
                line = line.strip()
                if not line or line[0] in ".;" or line.endswith(":"):
                    continue
                k = classify(line.split()[0])
                total[k] += 1
                cur[k] += 1
                if k == "BARRIER":
                    regions.append(cur)
                    cur = Counter()
            regions.append(cur)
            meta = text[end:end + 6000]
            vg = re.search(r"\.amdhsa_next_free_vgpr (\d+)", meta)
            sc = re.search(r"; ScratchSize: (\d+)", meta)
            print(f"{name[:90]}\n  total {dict(total)} vgpr={vg and vg.group(1)} scratch={sc and sc.group(1)}")
            for i, r in enumerate(regions):
                print(f"  region {i}: {dict(r)}")


if __name__ == "__main__":
    main()
