#!/bin/bash
# the round's measured evidence, on a GPU box, from the repository root:  profile_round.sh r03
# -> gpurun_out/<tag>/: bench lines (C3 default run, C5-shaped slice), rocprofv3 kernel stats, PMC passes + summary
set -u
tag=${1:-rXX}
out=$PWD/gpurun_out/$tag
mkdir -p $out
root=$PWD
python bench.py --steps 20 --warmup 5 > $out/${tag}_bench_c3.json 2> $out/bench_c3.err
echo "bench c3 done: $(python -c "import json;d=json.load(open('$out/${tag}_bench_c3.json'));print(d['ms_per_step'], d['value'])")"
python bench.py --workload c5s --steps 5 --warmup 2 --no-cpu-baseline --no-ingest > $out/${tag}_bench_c5s.json 2> $out/bench_c5s.err
echo "bench c5s done: $(python -c "import json;d=json.load(open('$out/${tag}_bench_c5s.json'));print(d['ms_per_step'], d['value'])")"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o p -- python3 $root/bench.py --no-cpu-baseline --no-full-pipeline --no-ingest --steps 3 --warmup 1 > /dev/null 2> $out/stats.err
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/${tag}_bench_c3_kernel_stats.csv
echo "kernel stats done"
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_$ctr -o p -- python3 $root/bench.py --no-cpu-baseline --no-full-pipeline --no-ingest --steps 1 --warmup 0 > /dev/null 2> $out/pmc_$ctr.err
  echo "pmc $ctr done"
done
cd $root
python msm_we_amd/csrc/tools/pmc_summary.py $out/${tag}_pmc_summary.csv $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE > /dev/null
rm -rf $out/stats $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
ls -la $out
