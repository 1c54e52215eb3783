# one or two host threads feeding the two queues of a latency-bound training pass (HAMSM_ENQUEUE_THREADS): parity, then C3 A/B
mkdir -p gpurun_out/r5o
timeout -k 10 900 python -m pytest tests/test_training_paths.py tests/test_reference_fuzz.py tests/test_oracle_golden.py tests/test_gpu_fullsize.py tests/test_ranked_training.py -m gpu -x -q > gpurun_out/r5o/pytest.log 2>&1; tail -1 gpurun_out/r5o/pytest.log
for v in 1 2 1 2 1 2; do
HAMSM_ENQUEUE_THREADS=$v HAMSM_DEBUG_ENQUEUE=1 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-full-pipeline --no-ingest > gpurun_out/r5o/b$v.json 2> gpurun_out/r5o/b$v.err
python -c "
import json;o=json.load(open('gpurun_out/r5o/b$v.json'));print('$v', o['ms_per_step'], o['stage_seconds_per_step']['step'])"
grep "enqueued" gpurun_out/r5o/b$v.err | tail -2
done
