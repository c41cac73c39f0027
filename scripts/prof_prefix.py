"""Per-request prefix (vision + splice + prefill) timeline from a rocprofv3 kernel-trace .db: kernels between the end of
the previous request's last decode step and this request's first decode GEMV, aggregated by name, plus the wall span."""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n[:100]


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute(f"select {namecol}, start, end from kernels order by start").fetchall()
    # a request's prefix starts at its patch-embedding im2col and ends at the first advance_kernel (sample_first)
    im = [i for i, r in enumerate(rows) if "im2col" in r[0]]
    if not im:
        print("no request found")
        return
    lo = max(0, im[-1] - 4)  # the torch.cat of image + depth and dtype copies just before it
    hi = next((i for i in range(im[-1], len(rows)) if "advance_kernel" in rows[i][0]), len(rows) - 1) + 1
    seg = rows[lo:hi]
    # cut at the first decode-step GEMV of the new request (embed_rows before it belongs to decode)
    span = (seg[-1][2] - seg[0][1]) / 1e6
    agg = {}
    for n, s, e in seg:
        a = agg.setdefault(n, [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e3
    busy = sum(a[1] for a in agg.values()) / 1e3
    print(f"# prefix of the last request: {len(seg)} kernels, wall {span:.3f} ms, GPU busy {busy:.3f} ms")
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"{a[0]:6d} {a[1] / 1e3:9.3f} ms {a[1] / a[0]:9.2f} us  {short(n)}")


if __name__ == "__main__":
    main(sys.argv[1])
