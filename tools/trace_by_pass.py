#!/usr/bin/env python
"""Per-pass kernel time from a rocprofv3 --kernel-trace rocpd database of tools/loop_drift.py: the dispatches are split into
passes at the long gaps where a new model is built, and for every pass the GPU-busy time per iteration, the wall time between the
pass's first and last kernel, and the average duration of the heaviest kernels are printed."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = next(t for t in tables if "kernel_dispatch" in t and "rocpd" in t)
sym = next((t for t in tables if "kernel_symbol" in t and "rocpd" in t), None)
cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
rows = db.execute(f"select kernel_id, start, end from {kd} order by start").fetchall()
names = {}
if sym:
    scols = [r[1] for r in db.execute(f"pragma table_info({sym})")]
    ncol = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else scols[1])
    for i, n in db.execute(f"select id, {ncol} from {sym}"):
        names[i] = n
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 80
passes, cur, last_end = [], [], None
for kid, s, e in rows:
    if last_end is not None and s - last_end > 30e6 and cur:           # > 30 ms without a kernel: between passes
        passes.append(cur)
        cur = []
    cur.append((kid, s, e))
    last_end = e
if cur:
    passes.append(cur)
for i, p in enumerate(passes):
    if len(p) < iters * 10:
        continue
    busy = sum(e - s for _, s, e in p) / 1e6
    wall = (p[-1][2] - p[0][1]) / 1e6
    per = {}
    for kid, s, e in p:
        n = names.get(kid, str(kid))
        k = n[n.find("k_"):n.find("k_") + 24] if "k_" in n else n[:40]
        a = per.setdefault(k, [0.0, 0])
        a[0] += (e - s) / 1e3
        a[1] += 1
    top = sorted(per.items(), key=lambda kv: -kv[1][0])[:6]
    print(f"segment {i}: {len(p)} kernels, busy {busy / iters:.3f} ms/iter, first-to-last {wall / iters:.3f} ms/iter; " +
          ", ".join(f"{k} {v[0] / v[1]:.1f}us" for k, v in top))
