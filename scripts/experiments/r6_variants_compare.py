"""per-shape launch times (serialized pass of bench.py, IEEE_PROFILE_DUMP) under several variants:
    python scripts/experiments/r6_variants_compare.py label1=a.csv label2=b.csv ...      (the first one is the base)"""
import csv
import sys
from collections import defaultdict


def load(path):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        acc[(r["kind"], " ".join(r["unit"].split(" ")[1:]))].append(float(r["us"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


tabs = [(a.split("=")[0], load(a.split("=")[1])) for a in sys.argv[1:]]
base = tabs[0][1]
print("%-6s %-30s %3s " % ("kind", "shape", "n") + " ".join("%9s" % t[0][:9] for t in tabs))
tot = [defaultdict(float) for _ in tabs]
for key in sorted(base, key=lambda k: (k[0], -base[k][0] * base[k][1])):
    if key[0] not in ("fwd", "dgrad", "wgrad"):
        continue
    row = []
    for i, (_, t) in enumerate(tabs):
        us = t.get(key, (float("nan"), 0))[0]
        row.append(us)
        tot[i][key[0]] += us * base[key][1]
    print("%-6s %-30s %3d " % (key[0], key[1], base[key][1]) + " ".join("%9.1f" % u for u in row))
for kind in ("fwd", "dgrad", "wgrad"):
    print("total %-6s (us over the dumped steps) " % kind + " ".join("%9.0f" % t[kind] for t in tot))
