#!/bin/bash
# issue-side counters of the ranked training E-step (k_step_assign_ranked) on a 40-iteration C5-shaped run:  esteppmc.sh [tag]   (repository root, GPU box)
tag=${1:-cur}
out=$PWD/gpurun_out/estep_pmc_$tag; rm -rf $out; mkdir -p $out; root=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_MFMA SQ_INSTS_LDS" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_VMEM SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --kernel-include-regex k_step_assign_ranked --output-format csv -d $out/p$i -o p -- \
    python3 $root/bench.py --workload ${HAMSM_PMC_WORKLOAD:-c5s} --iters 40 --steps 1 --warmup 1 --no-cpu-baseline --no-full-pipeline --no-ingest > /dev/null 2> $out/err_$i.txt
  echo "pass $i ($set) rc=$?"
done
cd $root
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for path in glob.glob("$out/p*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        if "k_step_assign_ranked" not in r["Kernel_Name"]: continue
        per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (_, c), v in per.items():
        a = acc[c]; a[0] += 1; a[1] += v
with open("$out/summary.txt", "w") as f:
    for c, (k, tot) in sorted(acc.items()):
        line = "%-34s n=%3d mean %16.1f" % (c, k, tot / k)
        print(line); f.write(line + "\n")
PY
