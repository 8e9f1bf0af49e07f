# usage (GPU box): scripts/step_gaps.sh [NAME=V ...]  -- how much of a training step the GPU is IDLE, and behind which kernels:
# rocprofv3 kernel trace of scripts/ab_step.py (8 warm-up + 40 steps), analysed over the last 20 steps: union of the kernel
# intervals (all streams) against wall time, the idle gaps by the kernel that ended before them, and kernel time that overlaps
# other kernels (the front end of the next minibatch, the text side stream).  Output: gpurun_out/step_gaps.txt
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sg
rocprofv3 --kernel-trace --output-format csv -d /tmp/sg -- python3 $root/scripts/ab_step.py "$@" > /tmp/sg.log 2>&1 || tail -5 /tmp/sg.log
tail -1 /tmp/sg.log
python3 - > $root/gpurun_out/step_gaps.txt <<'PY'
import csv, glob, re, collections
ev = []
for f in glob.glob("/tmp/sg/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "0"))))
ev.sort()
# the step boundary: the optimizer's last kernel recurs once per step; use the most frequent kernel name with ~48 occurrences
names = collections.Counter(e[2] for e in ev)
t_end = ev[-1][1]
# take the last 45 % of the trace's wall time that holds the two timed regions of 20 steps; simpler: the window of the last
# 20 occurrences of a once-per-step kernel
once = [n for n, c in names.items() if 48 <= c <= 60]
marker = None
for n in once:
    if "lsap" in n or "match_cost" in n:
        marker = n
        break
marker = marker or (once[0] if once else None)
if marker is None:
    print("no once-per-step kernel found", names.most_common(5))
    raise SystemExit
ts = [e[0] for e in ev if e[2] == marker]
per = len(ts) // 48 if len(ts) >= 48 else 1
ts = ts[::per]
t0, t1 = ts[-21], ts[-1]
win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
wall = (t1 - t0) / 20 / 1e3
busy, gaps, cur_end, last = 0, [], None, None
ksum = 0
for s, e, n, q in win:
    ksum += e - s
    if cur_end is None:
        cur_end, seg0, last = e, s, n
        continue
    if s > cur_end:
        busy += cur_end - seg0
        gaps.append((s - cur_end, last, n))
        seg0, cur_end, last = s, e, n
    elif e > cur_end:
        cur_end, last = e, n
busy += cur_end - seg0
print("marker kernel: %s" % marker[:80])
print("per step over 20 steps: wall %.1f us, GPU busy (union of kernels) %.1f us, idle %.1f us (%.1f %%), kernel time summed %.1f us (overlap %.1f us), %d launches" % (
    wall, busy / 20e3, wall - busy / 20e3, 100 * (1 - busy / 20e3 / wall), ksum / 20e3, (ksum - busy) / 20e3, len(win) // 20))
hist = collections.Counter()
for g, a, b in gaps:
    hist["<2us" if g < 2000 else "<5us" if g < 5000 else "<20us" if g < 20000 else "<100us" if g < 100000 else ">=100us"] += g
print("idle time per step by gap size: " + ", ".join("%s %.1f us" % (k, v / 20e3) for k, v in hist.items()))
by = collections.defaultdict(lambda: [0, 0])
short = lambda n: re.sub(r"\(anonymous namespace\)::|void |at::native::", "", n).split("(")[0][:70]
for g, a, b in gaps:
    k = (short(a), short(b))
    by[k][0] += 1
    by[k][1] += g
print("\nidle gaps by (kernel before -> kernel after), per step:")
for k, (n, g) in sorted(by.items(), key=lambda kv: -kv[1][1])[:50]:
    print("  %8.1f us  x%-5.1f  %s  ->  %s" % (g / 20e3, n / 20.0, k[0], k[1]))
PY
head -70 $root/gpurun_out/step_gaps.txt
