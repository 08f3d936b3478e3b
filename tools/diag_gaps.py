# Gaps between the kernels of one bench step, from a rocprofv3 --kernel-trace CSV (argument: the *_kernel_trace.csv).
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0] for r in rows]
gaps = collections.defaultdict(list)
for a, b, na, nb in zip(rows, rows[1:], names, names[1:]):
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1000.0
    if g < 200: gaps[na + " -> " + nb].append(g)
for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:12]:
    v = sorted(v)
    print("%-60s n %4d  gap us: median %.1f  p10 %.1f  p90 %.1f" % (k, len(v), v[len(v) // 2], v[len(v) // 10], v[9 * len(v) // 10]))
