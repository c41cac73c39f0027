"""one line per ubench_decode_mv run of a log (== label lines followed by the per-product lines)"""
import re, sys
rows, cur = [], None
for l in open(sys.argv[1]):
    if l.startswith('=='):
        cur = {'name': l.strip()[3:]}
        rows.append(cur)
    elif cur is not None:
        m = re.match(r'\s+(\S+)\s.*?([\d.]+) us', l)
        if m:
            cur[m.group(1)] = float(m.group(2))
        elif l.strip():
            cur.setdefault('msg', l.strip())
for r in rows:
    print(f"{r['name']:34s} qkv {r.get('qkv+norm', 0):6.2f}  o {r.get('o+res', 0):6.2f}  gu {r.get('gateup+norm+swiglu', 0):6.2f}  down {r.get('down+res', 0):6.2f}"
          f"  lm {r.get('lm_head+norm', 0):7.2f}  sum {r.get('sum', 0):6.2f} {r.get('msg', '')}")
