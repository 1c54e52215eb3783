#!/bin/bash
# kernel durations of the pipelined training pass (rocprofv3 kernel trace), by kernel and grid shape:  pipeprof2.sh [env assignments ...]
out=$PWD/gpurun_out/pipeprof; rm -rf $out; mkdir -p $out; root=$PWD
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out -o p -- python3 $root/bench.py ${BENCH_ARGS} --steps 3 --warmup 1 --no-cpu-baseline --no-full-pipeline --no-ingest > $out/bench.json 2> $out/err.txt
cd $root
python - <<PY
import csv, json, numpy as np
from collections import defaultdict
d = defaultdict(list)
for r in csv.DictReader(open("$out/p_kernel_trace.csv")):
    n = r["Kernel_Name"]
    if "k_step" not in n: continue
    d[(n.split("(")[0].replace("void ", "")[:34], int(r["Grid_Size_X"]) // 256, r["Grid_Size_Y"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
agg = defaultdict(list)
for (n, gx, gy), v in d.items(): agg[(n, gy)] += v
for k, v in sorted(agg.items()): print("%-36s y=%s  n=%5d  mean %6.1f  median %6.1f  min %6.1f" % (k[0], k[1], len(v), np.mean(v), np.median(v), np.min(v)))
b = json.load(open("$out/bench.json")); print("ms/step %.2f  training %.2f ms" % (b["ms_per_step"], b["stage_seconds_per_step"]["step"] * 1e3))
PY
