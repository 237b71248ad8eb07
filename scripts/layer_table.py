"""Per-layer roofline table of the conv kernels (profiles/rNN_layers.csv):
  python scripts/layer_table.py LAUNCHES.csv PMC_DIR OUT.csv [steps in LAUNCHES.csv]
LAUNCHES.csv = the per-launch HIP-event table of `IEEE_PROFILE_DUMP=... python bench.py` (one ordered stream: unit, kind, us,
gflop); PMC_DIR = rocprofv3 --pmc passes of the same command (scripts/pmc_passes.sh: fetch/ and write/ with FETCH_SIZE /
WRITE_SIZE per dispatch).  Launches and dispatches are matched by their order inside a training step (both are in enqueue
order); a weight gradient's row includes its slab-reduction dispatches.  Output per (layer shape, kind): launches per step,
mean us, GFLOP, TFLOP/s, fraction of the 2.5 PFLOP/s bf16 MFMA peak, HBM bytes per launch from the counters
(2 * FETCH_SIZE + WRITE_SIZE, KiB -> bytes), algorithmic bytes per launch (operands read once + result written once, 3
modalities; for dgrad launches including the operands of their fused BatchNorm-backward epilogue: mean over the launches
of the shape), achieved TB/s on the PMC bytes."""
import collections
import csv
import glob
import os
import re
import sys

launches, pmc_dir, out = sys.argv[1], sys.argv[2], sys.argv[3]
steps = int(sys.argv[4]) if len(sys.argv) > 4 else None
rows = list(csv.DictReader(open(launches)))
CONV = ("conv_gather_kernel", "conv3x3_patch_kernel", "stem_conv_kernel", "conv_wgrad_kernel", "conv3x3_wgrad_patch_kernel",
        "stem_wgrad_kernel")
REDUCE = ("wgrad_reduce_kernel", "wgrad_reduce_taps_kernel", "unpad_weight_grad_kernel")


def per_dispatch(counter):
    vals = {}
    for path in glob.glob(os.path.join(pmc_dir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                vals[(path, int(r["Dispatch_Id"]))] = (r["Kernel_Name"], float(r["Counter_Value"]))
    by_file = collections.defaultdict(list)
    for (path, did), v in vals.items():
        by_file[path].append((did, v))
    best = max(by_file.values(), key=len) if by_file else []
    return [v for _, v in sorted(best)]


def conv_sequence(disp):
    """[(kernel name, value incl. following reduce dispatches)] for the conv dispatches, split into steps at the layout kernel"""
    steps_, cur = [], None
    for name, val in disp:
        if "nchw_to_nhwc" in name:
            cur = []
            steps_.append(cur)
            continue
        if cur is None:
            continue
        if any(k in name for k in CONV):
            cur.append([name, val])
        elif any(k in name for k in REDUCE) and cur and "wgrad" in cur[-1][0]:
            cur[-1][1] += val
    return [s for s in steps_ if s]


fetch, write = conv_sequence(per_dispatch("FETCH_SIZE")), conv_sequence(per_dispatch("WRITE_SIZE"))
per_step = len(rows) // steps if steps else (len(fetch[-1]) if fetch else len(rows))
last = rows[-per_step:]
pmc = None
for f, w in zip(reversed(fetch), reversed(write)):          # a full step seen by both passes
    if len(f) == per_step and len(w) == per_step:
        pmc = [(2.0 * a[1] + b[1]) * 1024.0 for a, b in zip(f, w)]
        break
if pmc is None:
    print("warning: no PMC step with %d conv dispatches (fetch %s, write %s): the byte columns stay empty" %
          (per_step, [len(s) for s in fetch][-3:], [len(s) for s in write][-3:]))


def algorithmic_bytes(unit, kind, name=""):
    """operands read once + result written once, 3 modalities.  Round 4: a dgrad launch also carries the BatchNorm-backward
    sums of the unit in front of it in its epilogue, and those operands are part of what the launch MUST move: the previous
    unit's pre-BatchNorm output y (2 B per input element) for conv2 / conv3 / conv1, and at a block input (conv1 of every
    block but the first) the identity-branch gradient it adds (2 B; a quarter of that behind a stride-2 downsample branch,
    which hands its gradient back compact) and the packed ReLU bits of the previous block's output (1/8 B)."""
    m = re.search(r"(\d+)->(\d+) k(\d+) s(\d+) (\d+)x(\d+)", unit)
    ci, co, k, s, ho, wo = (int(x) for x in m.groups())
    B = 64
    out_e = B * ho * wo * co
    in_e = B * (ho * s) * (wo * s) * ci
    w_e = co * ci * k * k
    if kind == "wgrad":
        return 3 * (2 * (out_e + in_e) + 4 * w_e)
    extra = 0.0
    if kind == "dgrad" and ".layer" in name:
        leaf = name.rsplit(".", 1)[-1]
        blk = re.search(r"layer(\d)\.(\d+)\.", name)
        first_of_net = blk is not None and blk.group(1) == "1" and blk.group(2) == "0"
        if leaf in ("conv2", "conv3"):
            extra = 2.0 * in_e                                  # y of conv1 / conv2 for the fused BatchNorm-backward sums
        elif leaf == "conv1":
            has_ds = blk.group(2) == "0"
            compact = has_ds and blk.group(1) in ("2", "3")     # stride-2 stages: the downsample gradient comes back compact
            extra = 2.0 * in_e * (0.25 if compact else 1.0)     # the identity-branch gradient added in the epilogue
            if not first_of_net:
                extra += 2.0 * in_e + in_e / 8.0                # previous block's y3 + its packed ReLU bits
    return 3 * (2 * (out_e + in_e) + 2 * w_e + extra)


agg = collections.OrderedDict()
nrep = len(rows) // per_step
for i, r in enumerate(rows):
    name, shape = r["unit"].split(" ", 1)
    kind = r["kind"]
    a = agg.setdefault((shape, kind), dict(us=0.0, n=0, gflop=0.0, bytes=0.0, nb=0, alg=0.0))
    a["us"] += float(r["us"]); a["n"] += 1; a["gflop"] += float(r["gflop"]); a["alg"] += algorithmic_bytes(shape, kind, name)
for i, r in enumerate(last):
    if pmc is not None:
        a = agg[(r["unit"].split(" ", 1)[1], r["kind"])]
        a["bytes"] += pmc[i]; a["nb"] += 1
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["layer", "kind", "launches_per_step", "us", "gflop", "tflops", "frac_of_bf16_mfma_peak", "pmc_hbm_bytes",
                "algorithmic_bytes", "pmc_TBps"])
    for (shape, kind), a in sorted(agg.items(), key=lambda kv: -kv[1]["us"]):
        us, gf = a["us"] / a["n"], a["gflop"] / a["n"]
        pb = a["bytes"] / a["nb"] if a["nb"] else None
        w.writerow([shape, kind, a["n"] // nrep, "%.1f" % us, "%.2f" % gf, "%.0f" % (gf / us * 1e3), "%.3f" % (gf / us * 1e3 / 2500.0),
                    "%.0f" % pb if pb else "", "%.0f" % (a["alg"] / a["n"]), "%.2f" % (pb / us / 1e6) if pb else ""])
print("wrote", out, "(%d rows, %d launches per step, %d steps%s)" % (len(agg), per_step, nrep, "" if pmc is None else ", PMC bytes joined"))
