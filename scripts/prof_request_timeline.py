"""Timeline of the last request in a rocprofv3 kernel-trace .db: span from the end of the previous request's last kernel to this
request's first kernel (host time between requests), prefix span, decode span, and the largest idle gaps between consecutive kernels."""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute(f"select {namecol}, start, end from kernels order by start").fetchall()
    im = [i for i, r in enumerate(rows) if "im2col" in r[0]]
    if len(im) < 2:
        print("need two requests in the trace")
        return
    lo_prev, lo = max(0, im[-2] - 4), max(0, im[-1] - 4)
    first_adv = next(i for i in range(im[-1], len(rows)) if "advance_kernel" in rows[i][0])
    # the request ends at its last advance_kernel: the one before the largest idle gap that follows (the bench's roofline leg
    # and anything else after the request is cut off)
    adv = [i for i in range(first_adv, len(rows)) if "advance_kernel" in rows[i][0]]
    period = max(1, adv[1] - adv[0]) if len(adv) > 1 else 1
    last = adv[0]
    for a, b in zip(adv, adv[1:]):
        if b - a > 2 * period:
            break
        last = b
    rows = rows[:last + 1]
    print(f"previous request: first kernel -> this request's first kernel: {(rows[lo][1] - rows[lo_prev][1]) / 1e6:.3f} ms (= one request period)")
    print(f"idle between the previous request's last kernel and this request's first: {(rows[lo][1] - rows[lo - 1][2]) / 1e6:.3f} ms")
    print(f"prefix (first kernel .. first advance_kernel end): {(rows[first_adv][2] - rows[lo][1]) / 1e6:.3f} ms")
    print(f"decode (first advance_kernel end .. last kernel end): {(rows[-1][2] - rows[first_adv][2]) / 1e6:.3f} ms over {len(rows) - first_adv - 1} kernels")
    gaps = sorted(((rows[i + 1][1] - rows[i][2]) / 1e3, i) for i in range(lo, len(rows) - 1))[::-1][:14]
    print("largest idle gaps inside the request (us, after kernel -> before kernel):")
    for g, i in gaps:
        where = "prefix" if i < first_adv else "decode"
        print(f"  {g:9.1f}  [{where} #{i - lo}]  {rows[i][0][:60]}  ->  {rows[i + 1][0][:60]}")
    tot_prefix = sum((rows[i + 1][1] - rows[i][2]) for i in range(lo, first_adv)) / 1e6
    tot_dec = sum((rows[i + 1][1] - rows[i][2]) for i in range(first_adv, len(rows) - 1)) / 1e6
    print(f"sum of idle gaps: prefix {tot_prefix:.3f} ms, decode {tot_dec:.3f} ms")


if __name__ == "__main__":
    main(sys.argv[1])
