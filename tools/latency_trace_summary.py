import csv, glob, sys, collections
p = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(p)), key=lambda r: int(r["Start_Timestamp"]))
n = len(rows) // 6
last = rows[-n:]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last) * 1e-3
span = (int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) * 1e-3
print(f"last call: {n} kernels, busy {busy:.0f} us, span {span:.0f} us")
acc = collections.defaultdict(lambda: [0, 0.0])
for r in last:
    k = r["Kernel_Name"][:70]; acc[k][0] += 1; acc[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
for k, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"{c:5d} x {t / c:8.1f} us = {t:9.0f} us  {k}")
gaps = [(int(last[i + 1]["Start_Timestamp"]) - int(last[i]["End_Timestamp"])) * 1e-3 for i in range(n - 1)]
gaps.sort()
print("gaps: median %.1f us, p90 %.1f us, max %.1f us, sum %.0f us" % (gaps[len(gaps) // 2], gaps[int(len(gaps) * 0.9)], gaps[-1], sum(gaps)))
