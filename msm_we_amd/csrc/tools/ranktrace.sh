cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r4f && mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r4f
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4f/trace -o rk -- python3 msm_we_amd/csrc/tools/rankprof.py 60 > gpurun_out/r4f/rankprof.log 2>&1
echo rc=$?
ls gpurun_out/r4f/trace | head
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r4f/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
ev.sort()
E = [(s, e) for s, e, n in ev if "k_step_assign_ranked" in n]
M = [(s, e) for s, e, n in ev if "k_step_update_ranked" in n]
print("E launches", len(E), "M launches", len(M))
d = [ (e - s) / 1e3 for s, e in E]
import statistics
half = len(d)//2
for nm, x in (("E pass0", d[:half]), ("E pass1", d[half:])):
    print(nm, "mean %.1f median %.1f p90 %.1f max %.1f us" % (statistics.mean(x), statistics.median(x), sorted(x)[int(.9*len(x))], max(x)))
    print("   first 12:", " ".join("%.0f" % v for v in x[:12]))
    print("   last 12:", " ".join("%.0f" % v for v in x[-12:]))
dm = [(e - s)/1e3 for s, e in M]
print("M mean %.1f median %.1f max %.1f" % (statistics.mean(dm), statistics.median(dm), max(dm)))
by = collections.defaultdict(lambda: [0, 0.0])
for s, e, n in ev:
    k = n.split("(")[0][:60]
    by[k][0] += 1; by[k][1] += (e - s) / 1e3
for k, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:12]:
    print("%8d %10.1f us %8.2f avg  %s" % (c, t, t / c, k))
PY
