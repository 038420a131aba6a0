import re, sys
rows = {}
tile = None
for l in open(sys.argv[1]):
    m = re.match(r"=== tile (\d)", l)
    if m:
        tile = int(m.group(1)); continue
    p = l.split()
    if len(p) >= 5 and tile and p[0] not in ('layer', 'convs:', 'whole'):
        try: rows.setdefault(p[0], {})[tile] = (float(p[-2]), float(p[-1]))
        except ValueError: pass
print("%-30s" % "layer" + "".join("%16s" % ("tile%d ms/TF" % t) for t in range(1, 7)))
tot = 0
for k, v in rows.items():
    best = min(v, key=lambda t: v[t][0]); tot += v[best][0]
    print("%-30s" % k + "".join(("%9.3f/%5.1f%s" % (v[t][0], v[t][1], '*' if t == best else ' ')) if t in v else "%16s" % "-" for t in range(1, 7)))
print("sum of best: %.2f ms" % tot)
