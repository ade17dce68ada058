#!/usr/bin/env python
"""Where the multi-stream step's time goes, from a rocprofv3 --kernel-trace rocpd database of

    rocprofv3 --kernel-trace -d out -o t -- python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 --sustain-seconds 0

The dispatches of the timed steps are taken (the trace's last stretch of 500+ kernels without a gap longer than 2 ms) and the wall
time between the first and the last of them is split by WHAT was running at each instant: how many kernels at once, how
long at least one blend kernel (the VALU-bound ones) was on the GPU, how long only latency-bound kernels were, how long
nothing was.  Per kernel: launches, mean duration, mean gap to the previous kernel of the SAME queue (dependent launches of a
view: what a stream's chain loses between kernels).

    python tools/timeline.py <trace.db> [- [stretch index]]"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = next(t for t in tables if "kernel_dispatch" in t and "rocpd" in t)
sym = next((t for t in tables if "kernel_symbol" in t and "rocpd" in t), None)
cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
qcol = next((c for c in ("queue_id", "stream_id", "queue") if c in cols), None)
sel = f"select kernel_id, start, end{', ' + qcol if qcol else ''} from {kd} order by start"
rows = db.execute(sel).fetchall()
names = {}
if sym:
    scols = [r[1] for r in db.execute(f"pragma table_info({sym})")]
    ncol = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else scols[1])
    for i, n in db.execute(f"select id, {ncol} from {sym}"):
        names[i] = n


def short(kid):
    n = names.get(kid, str(kid))
    i = n.find("k_")
    return n[i:i + 26].split("(")[0] if i >= 0 else n[:30]


# the final stretch of the trace without a long gap = the timed steps
stretches, cur = [], [rows[0]]
for prev, r in zip(rows, rows[1:]):
    if r[1] - prev[2] >= 2e6:
        stretches.append(cur)
        cur = []
    cur.append(r)
stretches.append(cur)
# which stretch: the longest (the headline leg: warm-up + timed steps issued back to back), or the k-th from the end (argv[3])
which = int(sys.argv[3]) if len(sys.argv) > 3 else None
big = [s for s in stretches if len(s) >= 500]
seg = max(big, key=len) if which is None else big[which]
t0, t1 = seg[0][1], max(r[2] for r in seg)
wall = (t1 - t0) / 1e3
blend = lambda kid: "k_render_fwd" in names.get(kid, "") or "k_render_bwd" in names.get(kid, "")
events = []
for kid, s, e, *q in seg:
    events.append((s, 1, blend(kid)))
    events.append((e, -1, blend(kid)))
events.sort()
n_run = n_blend = 0
last = t0
by_conc, t_blend, t_small_only, t_idle = defaultdict(float), 0.0, 0.0, 0.0
for t, d, b in events:
    dt = (t - last) / 1e3
    by_conc[n_run] += dt
    if n_blend > 0:
        t_blend += dt
    elif n_run > 0:
        t_small_only += dt
    else:
        t_idle += dt
    n_run += d
    n_blend += d if b else 0
    last = t
views = sum(1 for r in seg if "k_render_fwd" in names.get(r[0], "")) or None       # one forward blend per view
print(f"{len(seg)} kernels over {wall:.1f} us" + (f" = {wall / views:.1f} us per view" if views else ""))
print(f"  at least one blend kernel running {100 * t_blend / wall:.1f} %, only other kernels {100 * t_small_only / wall:.1f} %, "
      f"nothing {100 * t_idle / wall:.1f} %")
print("  kernels running at once: " + ", ".join(f"{k}: {100 * v / wall:.1f} %" for k, v in sorted(by_conc.items())))
per, prev_end = defaultdict(lambda: [0.0, 0, 0.0, 0]), {}
for kid, s, e, *q in seg:
    k = short(kid)
    a = per[k]
    a[0] += (e - s) / 1e3
    a[1] += 1
    key = q[0] if q else 0
    if key in prev_end:
        a[2] += max(0.0, (s - prev_end[key]) / 1e3)
        a[3] += 1
    prev_end[key] = e
print("  kernel: launches, mean us, mean gap after the previous kernel of the same queue")
for k, a in sorted(per.items(), key=lambda kv: -kv[1][0]):
    print(f"    {k:28s} {a[1]:5d}  {a[0] / a[1]:7.1f}  {a[2] / max(a[3], 1):6.1f}")

# the gaps in front of a view's first kernel (the preprocess), per queue: where a stream waited between two views -- the host
# behind, or the step boundary (the streams are forked from and joined to the caller's stream once per step)
gaps, prev_end, queues = [], {}, set()
for kid, s, e, *q in seg:
    key = q[0] if q else 0
    if "k_preprocess" in names.get(kid, ""):
        queues.add(key)
        if key in prev_end:
            gaps.append(max(0.0, (s - prev_end[key]) / 1e3))
    prev_end[key] = e
if gaps:
    gaps.sort()
    big_g = [g for g in gaps if g > 100.0]
    print(f"  gap in front of a view's first kernel: {len(gaps)} views on {len(queues)} queues, median {gaps[len(gaps) // 2]:.1f} us, "
          f"mean {sum(gaps) / len(gaps):.1f} us; the {len(big_g)} gaps above 100 us (mean {sum(big_g) / max(len(big_g), 1):.0f} us) are "
          f"{100 * sum(big_g) / (max(len(queues), 1) * wall):.1f} % of every queue's wall time, all the others {100 * (sum(gaps) - sum(big_g)) / (max(len(queues), 1) * wall):.1f} %")
