#!/bin/bash
# memory-side counters of the training kernels, pipelined vs two-launch:  pipepmc.sh   (from the repository root, on a GPU box)
out=$PWD/gpurun_out/pipepmc; rm -rf $out; mkdir -p $out; root=$PWD
cd /tmp && export TMPDIR=/tmp
for mode in 1 0; do
  for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum"; do
    tag=$(echo $set | cut -d' ' -f1)
    HAMSM_TRAIN_AHEAD=$mode HAMSM_STEP_LANES=1 HAMSM_AHEAD_N0=200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/m${mode}_$tag -o p -- \
      python3 $root/bench.py --steps 1 --warmup 0 --iters 280 --no-cpu-baseline --no-full-pipeline --no-ingest > /dev/null 2> $out/err_${mode}_$tag.txt
  done
done
cd $root
python - <<PY
import csv, glob, collections
for mode in (1, 0):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for path in glob.glob("$out/m%d_*/**/*counter_collection.csv" % mode, recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(path)):
            n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
            if "k_step" not in n: continue
            per[(r["Dispatch_Id"], n + " gy=" + r.get("Grid_Size_Y", "?"), r["Counter_Name"])] += float(r["Counter_Value"])
        for (_, n, c), v in per.items():
            a = acc[(n, c)]; a[0] += 1; a[1] += v
    print("== pipelined" if mode else "== two launches per iteration")
    for (n, c), (k, tot) in sorted(acc.items()):
        print("  %-52s %-34s n=%5d mean %12.1f" % (n, c, k, tot / k))
PY
