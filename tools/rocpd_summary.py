"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average duration.

Usage: python tools/rocpd_summary.py gpurun_out/prof/x_results.db [--steps N] > profiles/<name>.txt
"""
import argparse
import re
import sqlite3


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"unsigned short", "bf16", name)
    return name if len(name) <= 110 else name[:107] + "..."


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--steps", type=int, default=1, help="divide totals by this many profiled steps")
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    con = sqlite3.connect(a.db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {namecol}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {namecol} order by sum(end-start) desc").fetchall()
    total = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace summary of {a.db}")
    print(f"# total kernel time {total/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches; per step (/{a.steps}): "
          f"{total/1e6/a.steps:.3f} ms")
    print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'pct':>6}  kernel")
    for name, n, tot, avg, mn, mx in rows[:a.top]:
        print(f"{n:7d} {tot/1e6:10.3f} {avg/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100.0*tot/total:6.2f}  {short(name)}")


if __name__ == "__main__":
    main()
