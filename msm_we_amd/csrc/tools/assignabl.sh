# kernel-only durations of k_assign_filter for the variants of tools/assignabl.py (7 launches per variant, in order)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/assignabl
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o a -- python3 $GRAFT_REPO_ROOT/msm_we_amd/csrc/tools/assignabl.py > $OUT/abl.log 2>&1
tail -5 $OUT/abl.log
python3 - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/assignabl/trace/**/*kernel_trace.csv", recursive=True)[0]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
d = [(e - s) / 1e3 for s, e, n in ev if "k_assign_filter" in n and "offsets" not in n]
x = [(e - s) / 1e3 for s, e, n in ev if "k_assign_sorted" in n]
names = ["base", "NOFRAG", "NORANK"]
for i, nm in enumerate(names):
    seg = d[7 * i: 7 * i + 7]
    if seg:
        print(nm, "k_assign_filter us:", " ".join("%.0f" % v for v in seg), " exact:", " ".join("%.0f" % v for v in x[7 * i: 7 * i + 7]))
PY
