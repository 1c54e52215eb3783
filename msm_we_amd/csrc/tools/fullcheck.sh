# the whole GPU suite, the experiments' suite, then the round's profiles:  fullcheck.sh r04
set -u
tag=${1:-rXX}
mkdir -p gpurun_out/$tag
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/$tag/pytest_gpu.log 2>&1; rc1=$?
tail -1 gpurun_out/$tag/pytest_gpu.log
[ $rc1 -ne 0 ] && exit $rc1
timeout -k 10 600 python -m pytest tests -m experiments -x -q > gpurun_out/$tag/pytest_exp.log 2>&1; rc2=$?
tail -1 gpurun_out/$tag/pytest_exp.log
[ $rc2 -ne 0 ] && exit $rc2
bash msm_we_amd/csrc/tools/profile_round.sh $tag
