#!/bin/bash
# where the GPU idles inside a bench step: kernel trace -> gaps over 15 us between consecutive kernels (repository root, GPU box)
out=$PWD/gpurun_out/gaptrace; rm -rf $out; mkdir -p $out; root=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/t -o p -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-full-pipeline --no-ingest > $out/bench.json 2> $out/err.txt
cd $root
python - <<PY
import csv, glob
rows = []
for path in glob.glob("$out/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]))
for path in glob.glob("$out/t/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "memcpy " + r.get("Direction", "")))
rows.sort()
# the last step: from the last k_prepass-or-first kernel after a long idle
t_end = rows[-1][1]
# find step boundaries: k_seed_gram launches (one per step)
seeds = [i for i, r in enumerate(rows) if r[2].startswith("k_seed_gram")]
i0 = seeds[-1]
# walk back to the previous flux end
while i0 > 0 and rows[i0][0] - rows[i0 - 1][1] < 200000: i0 -= 1
step = rows[i0:]
busy_end = step[0][1]; tot_gap = 0
print("last step: %d records, span %.3f ms" % (len(step), (step[-1][1] - step[0][0]) / 1e6))
for a, b in zip(step, step[1:]):
    busy_end = max(busy_end, a[1])
    gap = b[0] - busy_end
    if gap > 15000:
        tot_gap += gap
        print("  gap %7.1f us  after %-50s before %s" % (gap / 1e3, a[2], b[2]))
print("gaps over 15 us total %.3f ms" % (tot_gap / 1e6))
PY
