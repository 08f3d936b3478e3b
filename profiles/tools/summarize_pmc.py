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
json.dump(out, sys.stdout, indent=1)
