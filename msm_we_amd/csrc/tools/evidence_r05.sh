mkdir -p gpurun_out/ev
python msm_we_amd/csrc/tools/hostprof.py 10 2>&1 | tail -22 > gpurun_out/ev/r05_hostprof_c3.txt
HAMSM_PROF_WORKLOAD=c5s python msm_we_amd/csrc/tools/hostprof.py 3 2>&1 | tail -40 > gpurun_out/ev/r05_hostprof_c5s.txt
for v in "0 0" "1 0" "1 1" "0 0" "1 0" "1 1"; do
set -- $v
HAMSM_STEP_DIRECT_PLAIN=$1 HAMSM_STEP_LROWS=$2 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-full-pipeline --no-ingest > gpurun_out/ev/b.json 2>/dev/null
python -c "
import json;o=json.load(open('gpurun_out/ev/b.json'));r=o['roofline'];print('HAMSM_STEP_DIRECT_PLAIN=$1 HAMSM_STEP_LROWS=$2: step %.3f ms, training %.3f ms, E-step launch %.2f us (device stamps), M-step %.2f us (HIP events incl. boundary)' % (o['ms_per_step'], 1e3*o['stage_seconds_per_step']['step'], 1e3*r['avg_launch_ms'], 1e3*r['m_step_avg_launch_ms_hip_events']))"
done > gpurun_out/ev/r05_ab_direct_plain_c3.txt
for v in "2 1" "2 2" "3 2" "4 2" "2 1" "2 2"; do
set -- $v
HAMSM_STEP_LANES=$1 HAMSM_ENQUEUE_THREADS=$2 HAMSM_DEBUG_ENQUEUE=1 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-full-pipeline --no-ingest > gpurun_out/ev/b.json 2> gpurun_out/ev/b.err
python -c "
import json;o=json.load(open('gpurun_out/ev/b.json'));print('HAMSM_STEP_LANES=$1 HAMSM_ENQUEUE_THREADS=$2: step %.3f ms, training %.3f ms;' % (o['ms_per_step'], 1e3*o['stage_seconds_per_step']['step']), end=' ')"
grep "enqueued" gpurun_out/ev/b.err | tail -1
done > gpurun_out/ev/r05_ab_enqueue_c3.txt
cat gpurun_out/ev/r05_ab_direct_plain_c3.txt gpurun_out/ev/r05_ab_enqueue_c3.txt
