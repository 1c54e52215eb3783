# the exact pass behind the ranking pass: 64-frame workgroups on a capped grid against 512-frame workgroups on the full one
# (HAMSM_EXACT_CAPPED): parity, then the kernel's duration at C3 and C2 under the profiler, then the bench lines
mkdir -p gpurun_out/r5q
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py tests/test_assign_pair.py tests/test_gpu_fullsize.py tests/test_reference_fuzz.py tests/test_oracle_golden.py -m gpu -x -q > gpurun_out/r5q/pytest.log 2>&1; tail -1 gpurun_out/r5q/pytest.log
for v in 0 1; do
for wl in c3 c2; do
(cd /tmp && export TMPDIR=/tmp && HAMSM_EXACT_CAPPED=$v rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r5q/prof_$wl$v -o p -- python3 /root/repo/bench.py --workload $wl --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-full-pipeline --no-ingest > /dev/null 2>&1)
echo "capped=$v $wl: $(grep -E 'k_assign_sorted' gpurun_out/r5q/prof_$wl$v/p_kernel_stats.csv | cut -d, -f1-4 | cut -c1-40,90-)"
done
done
for v in 0 1 0 1; do
HAMSM_EXACT_CAPPED=$v python bench.py --steps 20 --warmup 5 --secondary c2 --no-cpu-baseline --no-full-pipeline --no-ingest > gpurun_out/r5q/b$v.json 2>/dev/null
python -c "
import json;o=json.load(open('gpurun_out/r5q/b$v.json'));print('capped=$v', o['ms_per_step'], o['stage_seconds_per_step']['assign'], o['secondary']['c2']['ms_per_step'])"
done
