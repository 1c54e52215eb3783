mkdir -p gpurun_out/r4q
timeout -k 10 300 python bench.py --workload c5s --steps 5 --warmup 2 --no-cpu-baseline --no-ingest --no-full-pipeline > gpurun_out/r4q/c5s.json 2> gpurun_out/r4q/c5s.err &&
HAMSM_DEBUG_REPORT=1 timeout -k 10 300 python bench.py --workload c5s --steps 1 --warmup 0 --no-cpu-baseline --no-ingest --no-full-pipeline > gpurun_out/r4q/c5s_dbg.json 2> gpurun_out/r4q/c5s_dbg.err &&
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ingest --no-full-pipeline > gpurun_out/r4q/c3.json 2> gpurun_out/r4q/c3.err &&
timeout -k 10 600 python -m pytest tests/test_ranked_training.py tests/test_training_paths.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r4q/pytest.log 2>&1
tail -3 gpurun_out/r4q/pytest.log
python msm_we_amd/csrc/tools/benchline.py gpurun_out/r4q/c5s.json gpurun_out/r4q/c3.json
grep -c "plan. run" gpurun_out/r4q/c5s_dbg.err
