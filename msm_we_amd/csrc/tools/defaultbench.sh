t0=$(date +%s.%N)
python bench.py > gpurun_out/default_bench.json 2> gpurun_out/default_bench.err; rc=$?
t1=$(date +%s.%N)
echo "rc=$rc wall $(python -c "print(round($t1-$t0,1))") s"
python msm_we_amd/csrc/tools/benchline.py gpurun_out/default_bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/default_bench.json"))
print({k: d[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "dtype", "vs_baseline", "higher_is_better", "data")})
print(d["config"])
print(d["roofline"]["bound"], d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["kind"], d["cpu_baseline"]["sample"])
PY
