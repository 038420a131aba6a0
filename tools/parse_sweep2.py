import re, sys
rows = {}; key = None
for l in open(sys.argv[1]):
    m = re.match(r"=== (.*)", l)
    if m: key = m.group(1).strip(); continue
    p = l.split()
    if len(p) >= 5 and key and p[0] not in ('layer', 'convs:', 'whole'):
        try: rows.setdefault(p[0], {})[key] = (float(p[-2]), float(p[-1]))
        except ValueError: pass
    if p and p[0] in ('convs:', 'whole'): print(key, l.strip())
keys = []
for v in rows.values():
    for k in v:
        if k not in keys: keys.append(k)
print("%-28s" % "layer" + "".join("%14s" % k[-12:] for k in keys))
tot = 0
for n, v in rows.items():
    best = min(v.values())[0]; tot += best
    print("%-28s" % n + "".join("%7.3f/%5.1f%s" % (v[k][0], v[k][1], '*' if v[k][0] == best else ' ') if k in v else "%14s" % "-" for k in keys))
print("sum of best %.2f ms" % tot)
