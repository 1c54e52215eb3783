#!/bin/bash
# steptimeline.sh [bench args]: kernel trace of a bench run; for the training kernels: durations, the gaps between consecutive launches
# of a queue, how much of the pass the two queues overlap.   Output under gpurun_out/timeline/.
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/timeline
rm -rf $out && mkdir -p $out
cd $root
rocprofv3 --kernel-trace --output-format csv -d $out/t -o p -- python3 bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-full-pipeline --no-ingest > $out/bench.json 2> $out/err.txt
python3 - <<'PY'
import csv, glob, os, statistics, collections
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
f = glob.glob(root + "/gpurun_out/timeline/t/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Queue_Id", r.get("Stream_Id", "?"))) for r in rows)
tr = [e for e in ev if "k_step_" in e[2]]
if not tr:
    raise SystemExit("no training kernels in the trace")
# the last training pass: the final contiguous burst of step kernels
bursts, cur = [], [tr[0]]
for e in tr[1:]:
    if e[0] - cur[-1][1] > 2_000_000:
        bursts.append(cur); cur = [e]
    else:
        cur.append(e)
bursts.append(cur)
b = max(bursts[-3:], key=len)
t0, t1 = b[0][0], max(e[1] for e in b)
print("pass: %d launches, %.2f ms wall" % (len(b), (t1 - t0) / 1e6))
byq = collections.defaultdict(list)
for e in b:
    byq[e[3]].append(e)
for q, es in byq.items():
    busy = sum(e[1] - e[0] for e in es)
    gaps = [es[i + 1][0] - es[i][1] for i in range(len(es) - 1)]
    print("queue %s: %d launches, busy %.2f ms, gaps: mean %.2f us, median %.2f, p90 %.2f, total %.2f ms" % (
        q, len(es), busy / 1e6, statistics.mean(gaps) / 1e3, statistics.median(gaps) / 1e3, sorted(gaps)[int(.9 * len(gaps))] / 1e3, sum(gaps) / 1e6))
    byk = collections.defaultdict(list)
    for e in es:
        byk[e[2][:48]].append((e[1] - e[0]) / 1e3)
    for k, v in byk.items():
        print("     %-50s n=%5d mean %7.2f us median %7.2f p90 %7.2f" % (k, len(v), statistics.mean(v), statistics.median(v), sorted(v)[int(.9 * len(v))]))
# overlap: time with >= 1 and with >= 2 kernels running
pts = sorted([(e[0], 1) for e in b] + [(e[1], -1) for e in b])
n = 0; last = pts[0][0]; t_ge1 = t_ge2 = 0
for t, d in pts:
    if n >= 1: t_ge1 += t - last
    if n >= 2: t_ge2 += t - last
    n += d; last = t
print("some kernel running %.2f ms, two or more %.2f ms, idle %.2f ms" % (t_ge1 / 1e6, t_ge2 / 1e6, (t1 - t0 - t_ge1) / 1e6))
PY
