"""Per-kernel sums of PMC counters from a rocprofv3 results .db (counter-collection pass)."""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"^void ", "", n)[:90]


def main(path, names):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    view = "counters_collection" if "counters_collection" in tabs else None
    if view is None:
        print("tables/views:", tabs)
        return
    cols = [r[1] for r in c.execute(f"pragma table_info({view})")]
    kcol = "kernel_name" if "kernel_name" in cols else [x for x in cols if "kernel" in x and "name" in x][0]
    ccol = "counter_name" if "counter_name" in cols else [x for x in cols if "counter" in x and "name" in x][0]
    vcol = "value" if "value" in cols else [x for x in cols if "value" in x][0]
    dcol = "dispatch_id" if "dispatch_id" in cols else None
    agg = {}
    for k, cn, v, d in c.execute(f"select {kcol}, {ccol}, {vcol}, {dcol or 0} from {view}"):
        if cn not in names:
            continue
        a = agg.setdefault(k, {})
        e = a.setdefault(cn, [0.0, set()])
        e[0] += float(v)
        e[1].add(d)
    print("# per-kernel counter sums over all dispatches (value summed over XCDs/SEs as rocprofv3 reports it)")
    print(f"{'dispatches':>10}  " + "  ".join(f"{n:>26}" for n in names) + "  kernel")
    for k, a in sorted(agg.items(), key=lambda kv: -max(v[0] for v in kv[1].values())):
        nd = max(len(v[1]) for v in a.values())
        print(f"{nd:10d}  " + "  ".join(f"{a.get(n, [0.0])[0]:26.1f}" for n in names) + "  " + short(k))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
