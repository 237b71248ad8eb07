"""per-shape launch times (serialized pass of bench.py, IEEE_PROFILE_DUMP) of the production library against the
measurement build whose conv epilogue is compiled out (scripts/experiments/libieee_noepi.so, -DIEEE_DBG_NOEPI):
    python scripts/experiments/r6_noepi_compare.py base.csv noepi.csv
The difference is the UPPER BOUND of what overlapping a tile's epilogue with the next tile's operand fetch can give."""
import csv
import sys
from collections import defaultdict


def load(path):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        shape = " ".join(r["unit"].split(" ")[1:])
        acc[(shape, r["kind"])].append(float(r["us"]))
    return acc


a, b = load(sys.argv[1]), load(sys.argv[2])
rows, tot = [], defaultdict(lambda: [0.0, 0.0])
for key in a:
    if key not in b or key[1] not in ("fwd", "dgrad"):
        continue
    n = len(a[key])
    ta, tb = sum(a[key]) / n, sum(b[key]) / len(b[key])
    rows.append((key[1], key[0], n, ta, tb))
    tot[key[1]][0] += sum(a[key]); tot[key[1]][1] += sum(b[key])
print("%-6s %-32s %3s %9s %9s %7s" % ("kind", "shape", "n", "us", "no-epi us", "share"))
for kind, shape, n, ta, tb in sorted(rows, key=lambda r: (r[0], -(r[3] - r[4]) * r[2])):
    print("%-6s %-32s %3d %9.1f %9.1f %6.0f%%" % (kind, shape, n, ta, tb, 100 * (ta - tb) / ta))
for kind, (ta, tb) in tot.items():
    print("total %-6s: %.0f us per pass with the epilogue, %.0f without (%.0f us = %.0f %%)" % (kind, ta, tb, ta - tb, 100 * (ta - tb) / ta))
