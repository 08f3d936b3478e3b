"""Average every collected counter per kernel launch: reads <dir>/pmc_*/**/_counter_collection.csv."""
import csv, glob, json, os, sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in glob.glob(os.path.join(root, "pmc_*", "*", "*_counter_collection.csv")):
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"].split("(")[0]
            a = acc[k][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
out = {k: {c: round(v[0] / v[1]) for c, v in sorted(cs.items())} for k, cs in sorted(acc.items()) if k.startswith("cimg_")}
# the kernel sources these counters belong to (bench.py reports the traffic only for the same sources)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
try:
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_hash", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "bench.py"))
    src = open(spec.origin).read()
    ns = {}
    exec(compile(src[src.index("def kernel_source_hash"):src.index("def pmc_traffic")], "hash", "exec"), {"os": os, "ROOT": os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")}, ns)
    out["_source_hash"] = ns["kernel_source_hash"]()
except Exception as exc:                       # never lose a counter run over the bookkeeping
    out["_source_hash"] = "unknown: %s" % exc
json.dump(out, sys.stdout, indent=1)
