# A/B of the two deterministic flux paths (by rows / sorted) on the C3 bench: tests first, then stage times and kernel durations
mkdir -p gpurun_out/r5g
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "flux or full" > gpurun_out/r5g/pytest.log 2>&1; tail -2 gpurun_out/r5g/pytest.log
cd /tmp && export TMPDIR=/tmp
for fr in 1 0; do
HAMSM_FLUX_ROWS=$fr rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r5g/prof_fr$fr -o fr -- python3 /root/repo/bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-full-pipeline --no-ingest > /root/repo/gpurun_out/r5g/bench_fr$fr.json 2>/dev/null
python3 - <<PY
import csv, json
o=json.load(open('/root/repo/gpurun_out/r5g/bench_fr$fr.json')); print($fr, o['ms_per_step'], o['stage_seconds_per_step']['flux'])
rows=list(csv.DictReader(open('/root/repo/gpurun_out/r5g/prof_fr$fr/fr_kernel_trace.csv')))
for k in ('k_fr_row','k_fr_place','k_fr_count','k_fr_scan_t','k_flux_cells','k_flux_parts','k_flux_keys','k_flux_pack','onesweep'):
    d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows if k in r['Kernel_Name']]
    if d: print('  ', k, 'max', max(d), 'median', sorted(d)[len(d)//2], 'calls', len(d))
PY
done
