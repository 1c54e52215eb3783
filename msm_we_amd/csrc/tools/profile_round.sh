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
for wl in c3 c5s c2 c2_x64; do
  sfx=""; [ $wl != c3 ] && sfx="_$wl"
  wlarg=$wl; unset HAMSM_BENCH_C2_REPS
  if [ $wl = c2_x64 ]; then wlarg=c2; export HAMSM_BENCH_C2_REPS=64; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$wl -o p -- python3 $root/bench.py --workload $wlarg --no-cpu-baseline --no-full-pipeline --no-ingest --no-secondary --steps 3 --warmup 1 > /dev/null 2> $out/stats_$wl.err
  cp $(find $out/stats_$wl -name "*kernel_stats.csv" | head -1) $out/${tag}_bench_${wl}_kernel_stats.csv
  echo "kernel stats $wl done"
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_${wl}_$ctr -o p -- python3 $root/bench.py --workload $wlarg --no-cpu-baseline --no-full-pipeline --no-ingest --no-secondary --steps 1 --warmup 0 > /dev/null 2> $out/pmc_${wl}_$ctr.err
    echo "pmc $wl $ctr done"
  done
  (cd $root && python msm_we_amd/csrc/tools/pmc_summary.py $out/${tag}_pmc_summary$sfx.csv $out/pmc_${wl}_FETCH_SIZE $out/pmc_${wl}_WRITE_SIZE > /dev/null)
  rm -rf $out/stats_$wl $out/pmc_${wl}_FETCH_SIZE $out/pmc_${wl}_WRITE_SIZE
done
# the PMC summaries the bench lines quote as `traffic` are the ones just taken (the bench reads profiles/<tag>_pmc_summary*.csv)
cd $root
unset HAMSM_BENCH_C2_REPS
cp $out/${tag}_pmc_summary*.csv profiles/
# the default invocation: C3 (the headline) with C2, C2 x64 and C5 at full size under `secondary` (round 5)
python bench.py --steps 20 --warmup 5 > $out/${tag}_bench_default.json 2> $out/bench_default.err
echo "bench default done: $(python -c "import json;d=json.load(open('$out/${tag}_bench_default.json'));print(d['ms_per_step'], d['value'], {k: v['ms_per_step'] for k, v in d['secondary'].items()})")"
python bench.py --workload c5s --steps 5 --warmup 2 --no-cpu-baseline --no-ingest > $out/${tag}_bench_c5s.json 2> $out/bench_c5s.err
echo "bench c5s done: $(python -c "import json;d=json.load(open('$out/${tag}_bench_c5s.json'));print(d['ms_per_step'], d['value'])")"
ls -la $out
