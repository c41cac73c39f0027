"""Summarise a rocprofv3 results .db (kernel trace) into a per-kernel stats table (markdown-ish text)."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:110]


def main(path, top=40):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute(f"select {namecol}, start, end from kernels").fetchall()
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(n, [0, 0.0, 1e30, 0.0])
        d = (e - s) / 1e3
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    print(f"# kernels: {len(rows)} dispatches, total GPU kernel time {total / 1e3:.3f} ms")
    print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'pct':>6}  kernel")
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{a[0]:7d} {a[1] / 1e3:10.3f} {a[1] / a[0]:9.2f} {a[2]:9.2f} {a[3]:9.2f} {100 * a[1] / total:6.2f}  {short(n)}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
