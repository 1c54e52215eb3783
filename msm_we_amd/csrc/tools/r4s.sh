mkdir -p gpurun_out/r4s
HAMSM_DEBUG_ENQUEUE=1 HAMSM_DEBUG_REPORT=1 timeout -k 10 600 python bench.py --workload c5 --steps 1 --warmup 1 --no-cpu-baseline --no-ingest --no-full-pipeline > gpurun_out/r4s/c5_dbg.json 2> gpurun_out/r4s/c5_dbg.err
grep "plan. run\|hamsm\]" gpurun_out/r4s/c5_dbg.err | cut -c1-230 | tail -30
