# D > 128: the parity tests that touch the generic kernels, then the wide-feature bench with and without the ranking pass
mkdir -p gpurun_out/wide
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "wide or matches_oracle or random_shapes or 128 or generic" > gpurun_out/wide/pytest.log 2>&1; tail -1 gpurun_out/wide/pytest.log
timeout -k 10 300 python msm_we_amd/csrc/tools/widebench.py > gpurun_out/wide/wide_filter.log 2>&1; tail -3 gpurun_out/wide/wide_filter.log
HAMSM_ASSIGN_FILTER=0 timeout -k 10 300 python msm_we_amd/csrc/tools/widebench.py > gpurun_out/wide/wide_exact.log 2>&1; tail -3 gpurun_out/wide/wide_exact.log
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/wide/stats
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/wide/stats -o p -- python3 $GRAFT_REPO_ROOT/msm_we_amd/csrc/tools/widebench.py > /dev/null 2>&1
grep -i "assign_generic\|k_pack_gen\|k_gen_norms\|k_sort\|filter_generic" $(find $GRAFT_REPO_ROOT/gpurun_out/wide/stats -name "*kernel_stats.csv" | head -1) | cut -c1-150
