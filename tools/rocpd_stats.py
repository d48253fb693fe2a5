"""Summarise a rocprofv3 rocpd SQLite result (kernel-trace) into a per-kernel stats table (like --stats CSV).
usage: python tools/rocpd_stats.py <results.db> [out.md]"""
import re
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"\(.*$", "", name)
    return name[:110]


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, start, end from kernels").fetchall() if {"name", "start", "end"} <= set(cols) else None
    if rows is None:
        raise SystemExit(f"unexpected schema: {cols}")
    agg = {}
    t0, t1 = min(r[1] for r in rows), max(r[2] for r in rows)
    for name, s, e in rows:
        a = agg.setdefault(short(name), [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"| {k} | {a[0]} | {a[1] / 1e6:.3f} | {a[1] / a[0] / 1e3:.2f} | {a[2] / 1e3:.2f} | {a[3] / 1e3:.2f} | {100 * a[1] / total:.1f} |")
    # union of the kernels' [start, end) intervals = time the GPU ran at least one kernel; 1 - union/span = idle (host or dependency gaps)
    ev = sorted((s, e) for _, s, e in rows)
    busy, cs, ce = 0, ev[0][0], ev[0][1]
    for s_, e_ in ev[1:]:
        if s_ > ce:
            busy += ce - cs
            cs, ce = s_, e_
        else:
            ce = max(ce, e_)
    busy += ce - cs
    lines.append(f"\nkernel time total {total / 1e6:.2f} ms over wall span {(t1 - t0) / 1e6:.2f} ms ({len(rows)} dispatches); "
                 f"GPU busy (union of kernel intervals) {busy / 1e6:.2f} ms = {100.0 * busy / (t1 - t0):.1f} % of the span, "
                 f"mean concurrency while busy {total / busy:.2f}")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(*sys.argv[1:3])
