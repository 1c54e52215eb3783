# queues of a latency-bound training pass and the host threads that feed them (HAMSM_STEP_LANES, HAMSM_ENQUEUE_THREADS): C3 A/B
mkdir -p gpurun_out/r5o
for v in "2 2" "4 2" "3 2" "2 2" "4 2" "4 1"; do
set -- $v
HAMSM_STEP_LANES=$1 HAMSM_ENQUEUE_THREADS=$2 HAMSM_DEBUG_ENQUEUE=1 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-full-pipeline --no-ingest > gpurun_out/r5o/b$1$2.json 2> gpurun_out/r5o/b$1$2.err
python -c "
import json;o=json.load(open('gpurun_out/r5o/b$1$2.json'));print('lanes $1 threads $2:', o['ms_per_step'], o['stage_seconds_per_step']['step'])"
grep "enqueued" gpurun_out/r5o/b$1$2.err | tail -1
done
