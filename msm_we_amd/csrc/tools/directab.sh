# A/B of the f64-MFMA training kernels: with the f64 batch copy (HAMSM_STEP_DIRECT_PLAIN=0), member rows from the frames (1), and
# the row list in LDS on top (HAMSM_STEP_LROWS): parity first, then C3
mkdir -p gpurun_out/r5k
HAMSM_STEP_DIRECT_PLAIN=1 HAMSM_STEP_LROWS=1 timeout -k 10 900 python -m pytest tests/test_training_paths.py tests/test_gpu_parity.py tests/test_reference_fuzz.py tests/test_oracle_golden.py tests/test_gpu_fullsize.py tests/test_ranked_training.py -m gpu -x -q > gpurun_out/r5k/pytest.log 2>&1; tail -1 gpurun_out/r5k/pytest.log
for v in "0 0" "1 0" "1 1" "0 0" "1 0" "1 1"; do
set -- $v
HAMSM_STEP_DIRECT_PLAIN=$1 HAMSM_STEP_LROWS=$2 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-full-pipeline --no-ingest > gpurun_out/r5k/bench_dp$1$2.json 2>/dev/null
python -c "
import json;o=json.load(open('gpurun_out/r5k/bench_dp$1$2.json'));r=o['roofline'];print('$v', o['ms_per_step'], o['stage_seconds_per_step']['step'], r['avg_launch_ms'], r['avg_launch_ms_hip_events'], r['m_step_avg_launch_ms_hip_events'])"
done
