# kernel trace of one C5-shaped bench step: E / M launch durations by grid size, per-queue busy time and overlap of the training launches
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/c5trace
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o c5 -- python3 bench.py --workload ${HAMSM_TRACE_WORKLOAD:-c5s} --steps 2 --warmup 1 --no-cpu-baseline --no-ingest --no-full-pipeline > $OUT/bench.json 2> $OUT/bench.err
echo rc=$?
python3 - <<'PY'
import csv, glob, collections, statistics, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/c5trace"
f = glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
cols = rows[0].keys()
gx = "Grid_Size_X" if "Grid_Size_X" in cols else None
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r[gx]) if gx else 0, int(r.get("Grid_Size_Y", 0) or 0), r.get("Queue_Id", "")) for r in rows]
ev.sort()
for key in ("k_step_assign_ranked", "k_step_update_ranked", "k_step_reassign_ranked", "k_rank_pack", "k_step_check"):
    by = collections.defaultdict(list)
    for s, e, n, g, gy, q in ev:
        if key in n:
            by[(g, gy)].append((e - s) / 1e3)
    for k, v in sorted(by.items()):
        print("%-24s grid %-14s n %5d  mean %6.1f  median %6.1f  p90 %6.1f  min %5.1f us" % (key, k, len(v), statistics.mean(v), statistics.median(v), sorted(v)[int(.9 * len(v))], min(v)))
tr = [(s, e, q) for s, e, n, g, gy, q in ev if "k_step_" in n]
if tr:
    qs = sorted({q for _, _, q in tr})
    t0, t1 = min(s for s, _, _ in tr), max(e for _, e, _ in tr)
    for q in qs:
        busy = sum(e - s for s, e, qq in tr if qq == q)
        print("queue", q, "launches", sum(1 for x in tr if x[2] == q), "busy %.2f ms" % (busy / 1e6))
    print("span of all training launches %.2f ms (3 passes)" % ((t1 - t0) / 1e6))
PY
