#!/bin/bash
# the round's measured evidence, on a GPU box, from the repository root:  profile_round.sh r04
# -> gpurun_out/<tag>/: bench lines (C3 default run, C5 at its stated size, the C5-shaped slice), rocprofv3 kernel stats and PMC
#    passes + summaries for C3 and for the C5-shaped slice (same per-iteration launches as C5)
set -u
tag=${1:-rXX}
out=$PWD/gpurun_out/$tag
mkdir -p $out
root=$PWD
cd /tmp && export TMPDIR=/tmp
for wl in c3 c5s; do
  sfx=""; [ $wl = c5s ] && sfx="_c5s"
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$wl -o p -- python3 $root/bench.py --workload $wl --no-cpu-baseline --no-full-pipeline --no-ingest --steps 3 --warmup 1 > /dev/null 2> $out/stats_$wl.err
  cp $(find $out/stats_$wl -name "*kernel_stats.csv" | head -1) $out/${tag}_bench_${wl}_kernel_stats.csv
  echo "kernel stats $wl done"
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_${wl}_$ctr -o p -- python3 $root/bench.py --workload $wl --no-cpu-baseline --no-full-pipeline --no-ingest --steps 1 --warmup 0 > /dev/null 2> $out/pmc_${wl}_$ctr.err
    echo "pmc $wl $ctr done"
  done
  (cd $root && python msm_we_amd/csrc/tools/pmc_summary.py $out/${tag}_pmc_summary$sfx.csv $out/pmc_${wl}_FETCH_SIZE $out/pmc_${wl}_WRITE_SIZE > /dev/null)
  rm -rf $out/stats_$wl $out/pmc_${wl}_FETCH_SIZE $out/pmc_${wl}_WRITE_SIZE
done
# the PMC summaries the bench lines quote as `traffic` are the ones just taken (the bench reads profiles/<tag>_pmc_summary*.csv)
cd $root
cp $out/${tag}_pmc_summary.csv $out/${tag}_pmc_summary_c5s.csv profiles/
python bench.py --steps 20 --warmup 5 > $out/${tag}_bench_c3.json 2> $out/bench_c3.err
echo "bench c3 done: $(python -c "import json;d=json.load(open('$out/${tag}_bench_c3.json'));print(d['ms_per_step'], d['value'])")"
python bench.py --workload c5s --steps 5 --warmup 2 --no-cpu-baseline --no-ingest > $out/${tag}_bench_c5s.json 2> $out/bench_c5s.err
echo "bench c5s done: $(python -c "import json;d=json.load(open('$out/${tag}_bench_c5s.json'));print(d['ms_per_step'], d['value'])")"
python bench.py --workload c5 --steps 3 --warmup 1 > $out/${tag}_bench_c5.json 2> $out/bench_c5.err
echo "bench c5 done: $(python -c "import json;d=json.load(open('$out/${tag}_bench_c5.json'));print(d['ms_per_step'], d['value'])")"
ls -la $out
