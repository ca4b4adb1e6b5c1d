"""Timeline statistics of a rocprofv3 (rocpd sqlite) kernel trace: busy union, idle gaps, two-stream overlap.
usage: python tools/rocpd_timeline.py x_results.db [--skip-first-ms N]"""
import argparse
import sqlite3

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--last-ms", type=float, default=0.0, help="only look at the last N ms of the trace")
a = ap.parse_args()
con = sqlite3.connect(a.db)
cur = con.cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type='table' or type='view'")]
kt = [t for t in tables if t.startswith("rocpd_kernel_dispatch")] or [t for t in tables if "kernel" in t]
cols = [r[1] for r in cur.execute(f"pragma table_info({kt[0]})")]
print("# table", kt[0], "cols", cols)
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = cur.execute(f"select start, end{', ' + qcol if qcol else ''} from {kt[0]} order by start").fetchall()
t1 = max(r[1] for r in rows)
if a.last_ms > 0:
    rows = [r for r in rows if r[0] >= t1 - a.last_ms * 1e6]
t0 = rows[0][0]
# union busy time and gaps
busy, gaps, cur_end = 0, [], rows[0][0]
cs = rows[0][0]
for r in rows:
    s, e = r[0], r[1]
    if s > cur_end:
        busy += cur_end - cs
        gaps.append((s - cur_end, cur_end - t0))
        cs = s
    cur_end = max(cur_end, e)
busy += cur_end - cs
tot = t1 - t0
ksum = sum(r[1] - r[0] for r in rows)
print(f"span {tot / 1e6:.2f} ms | union busy {busy / 1e6:.2f} ms ({100 * busy / tot:.1f} %) | sum of kernel durations {ksum / 1e6:.2f} ms "
      f"| mean concurrency while busy {ksum / busy:.2f}")
gaps.sort(reverse=True)
print(f"idle total {(tot - busy) / 1e6:.2f} ms in {len(gaps)} gaps; >20us: {sum(1 for g in gaps if g[0] > 20000)} gaps, "
      f"{sum(g[0] for g in gaps if g[0] > 20000) / 1e6:.2f} ms; largest (ms @ ms): "
      + ", ".join(f"{g[0] / 1e6:.2f}@{g[1] / 1e6:.0f}" for g in gaps[:12]))
if qcol:
    per = {}
    for s, e, q in rows:
        per.setdefault(q, [0, 0])
        per[q][0] += e - s
        per[q][1] += 1
    for q, (d, n) in sorted(per.items(), key=lambda kv: -kv[1][0]):
        print(f"  {qcol} {q}: {n} kernels, {d / 1e6:.2f} ms")

# kernels around the largest gaps
try:
    sym = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
    names = dict(cur.execute(f"select id, kernel_name from {sym}").fetchall())
    krows = cur.execute(f"select start, end, kernel_id, queue_id from {kt[0]} order by start").fetchall()
    if a.last_ms > 0:
        krows = [r for r in krows if r[0] >= t1 - a.last_ms * 1e6]
    ends = sorted(krows, key=lambda r: r[1])
    for gap, at in gaps[:4]:
        gs = t0 + at
        before = [r for r in krows if r[1] <= gs + 1000][-3:]
        after = [r for r in krows if r[0] >= gs + gap - 1000][:3]
        print(f"gap {gap / 1e6:.2f} ms:")
        for r in before:
            print(f"   before q{r[3]} {names.get(r[2], '?')[:90]}")
        for r in after:
            print(f"   after  q{r[3]} {names.get(r[2], '?')[:90]}")
except Exception as ex:  # noqa: BLE001
    print("# (no kernel names:", ex, ")")

# per-queue kernel classes
try:
    import collections
    import re
    agg = collections.defaultdict(lambda: [0, 0])
    for st, en, kid, q in krows:
        nm = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", names.get(kid, "?"))[:60]
        agg[(q, nm)][0] += en - st
        agg[(q, nm)][1] += 1
    for (q, nm), (d, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
        print(f"  q{q} {d / 1e6:8.2f} ms {n:5d}  {nm}")
except Exception as ex:  # noqa: BLE001
    print("# (no per-queue classes:", ex, ")")
