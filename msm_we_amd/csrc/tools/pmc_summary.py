"""Summarise rocprofv3 --pmc passes (one counter per pass, as MI355X_MICROARCH.md prescribes) into the table bench.py reads:
    python pmc_summary.py out.csv pass_dir [pass_dir ...]
Each pass_dir holds a *_counter_collection.csv; rows are averaged per (kernel, counter) over dispatches."""
import csv, glob, os, sys
from collections import defaultdict

out, dirs = sys.argv[1], sys.argv[2:]
acc = defaultdict(lambda: [0, 0.0])
for d in dirs:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per_dispatch = defaultdict(float)
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"].split("(")[0]
            per_dispatch[(row["Dispatch_Id"], name, row["Counter_Name"])] += float(row["Counter_Value"])
        for (_, name, counter), v in per_dispatch.items():
            a = acc[(name, counter)]
            a[0] += 1
            a[1] += v
with open(out, "w") as f:
    f.write("kernel,counter,dispatches,mean_value_per_dispatch\n")
    for (name, counter), (n, tot) in sorted(acc.items(), key=lambda kv: (kv[0][1], -kv[1][1])):
        f.write(f'"{name}",{counter},{n},{tot / n:.3f}\n')
print(open(out).read())
