# a round of checks after a kernel change: the parity files that exercise it, then the three bench shapes (digest lines)
mkdir -p gpurun_out/quick
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_ranked_training.py -m gpu -x -q > gpurun_out/quick/pytest.log 2>&1; rc=$?
tail -2 gpurun_out/quick/pytest.log | cut -c1-300
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --workload c5s --steps 5 --warmup 2 --no-cpu-baseline --no-ingest --no-full-pipeline > gpurun_out/quick/c5s.json 2> gpurun_out/quick/c5s.err &&
timeout -k 10 600 python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline --no-ingest --no-full-pipeline > gpurun_out/quick/c5.json 2> gpurun_out/quick/c5.err &&
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ingest --no-full-pipeline > gpurun_out/quick/c3.json 2> gpurun_out/quick/c3.err
python msm_we_amd/csrc/tools/benchline.py gpurun_out/quick/c5s.json gpurun_out/quick/c5.json gpurun_out/quick/c3.json
