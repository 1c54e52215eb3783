#!/bin/bash
# look-ahead blocks: bench step time over (first block after N0 iterations, most iterations per block) settings
# usage: aheadsweep.sh "N0 G [ALPHA]" ...     (run from the repository root on a GPU box)
mkdir -p gpurun_out/ahead
for cfg in "$@"; do
  set -- $cfg
  export HAMSM_AHEAD_N0=$1 HAMSM_AHEAD_G=$2 HAMSM_AHEAD_ALPHA=${3:-1.5}
  tag="n0_$1_g_$2_a_${3:-1.5}"
  HAMSM_DEBUG_AHEAD=1 python bench.py --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline --no-full-pipeline --no-ingest ${BENCH_ARGS} \
    > gpurun_out/ahead/$tag.json 2> gpurun_out/ahead/$tag.err
  echo "== N0=$1 G=$2 alpha=${3:-1.5}: blocks given up $(grep -c 'given up' gpurun_out/ahead/$tag.err)"
  grep "given up" gpurun_out/ahead/$tag.err | sed 's/.*position \([0-9]*\).*unserved flags \([0-9]*\)/pos \1 flags \2/' | sort | uniq -c | sort -rn | head -4
  python - <<PY
import json
d = json.load(open("gpurun_out/ahead/$tag.json"))
print("   ms/step %.2f  training steps %.2f ms  value %.1f M frames/s" % (d["ms_per_step"], d["stage_seconds_per_step"]["step"] * 1e3, d["value"] / 1e6))
PY
done
