"""Summarise rocprofv3 --pmc passes (counter_collection.csv files) per kernel family into one JSON for profiles/.

  python scripts/pmc_summary.py OUT.json DIR [DIR ...]      (every *_counter_collection.csv below the DIRs is read)

Each pass was collected on its own (`rocprofv3 --pmc <counters> --output-format csv -d DIR -- python3 bench.py ...`, no
trace domains beside it).  Per family: launches seen, the mean of every counter per launch, and the derived figures
  hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024     gfx950: FETCH_SIZE counts 64 B per 128-B request on
                                                                  wide coalesced reads (MI355X_MICROARCH.md, HBM)
  l2_hit_rate          = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
  mfma_busy            = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 4 SIMDs * 256 CUs)   share of MFMA-pipe cycles
  wait_share           = SQ_WAIT_ANY / SQ_WAVE_CYCLES          wave-cycles parked in s_waitcnt / barriers
  issue_stall_share    = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
  active_share         = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES"""
import collections
import csv
import glob
import json
import os
import sys

FAMILIES = (
    ("conv_patch", "conv3x3_patch_kernel"), ("conv_wgrad_patch", "conv3x3_wgrad_patch_kernel"), ("stem_fwd", "stem_conv_kernel"),
    ("stem_wgrad", "stem_wgrad_kernel"), ("conv_gather", "conv_gather_kernel"), ("conv_wgrad", "conv_wgrad_kernel"), ("wgrad_reduce", "wgrad_reduce_kernel"),
    ("bn_bwd_apply", "bn_bwd_apply"), ("bn_apply", "bn_apply"), ("bn_bwd_reduce", "bn_bwd_reduce_kernel"),
    ("bn_finalize", "bn_finalize_kernel"), ("bn_finalize", "bn_bwd_finalize_kernel"), ("distmat_f16split", "distmat_kernel<bool _Accum"),
    ("distmat_f16split", "distmat_kernelIDF16bLb1"), ("distmat_bf16", "distmat_kernelIDF16b"), ("distmat_fp32", "distmat_kernel<float"),
    ("distmat_fp32", "distmat_kernelIf"), ("distmat", "distmat_kernel"),
    ("rank_query_fast", "rank_query_fast_kernel"), ("rank_finalize", "rank_finalize_kernel"), ("split_rows", "split_rows"),
    ("sgd", "sgd_nesterov_kernel"), ("pack", "pack_all_kernel"), ("cim", "cim_"), ("sgemm", "sgemm_"),
)


def family(name):
    for fam, key in FAMILIES:
        if key in name:
            return fam
    return "other"


BY_GRID = False


def main():
    global BY_GRID
    args = [a for a in sys.argv[1:] if a != "--by-grid"]
    BY_GRID = len(args) != len(sys.argv) - 1
    out_path, dirs = args[0], args[1:]
    sums = collections.defaultdict(lambda: collections.defaultdict(float))
    counts = collections.defaultdict(lambda: collections.defaultdict(int))
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    fam = family(row["Kernel_Name"])
                    c = row["Counter_Name"]
                    keys = [fam]
                    if BY_GRID and fam in ("conv_gather", "conv_wgrad"):       # one entry per launch geometry too
                        keys.append("%s/grid%s_lds%s_vgpr%s" % (fam, row["Grid_Size"], row["LDS_Block_Size"], row["VGPR_Count"]))
                    for k in keys:
                        sums[k][c] += float(row["Counter_Value"])
                        counts[k][c] += 1
    step_launches = 0
    step_fetch = step_write = 0.0
    fam_bytes = collections.defaultdict(float)
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    c = row["Counter_Name"]
                    if c == "FETCH_SIZE":
                        step_fetch += float(row["Counter_Value"])
                        fam_bytes[family(row["Kernel_Name"])] += 2.0 * float(row["Counter_Value"]) * 1024.0
                        if "margin3m_kernel" in row["Kernel_Name"]:
                            step_launches += 1        # one per train step
                    elif c == "WRITE_SIZE":
                        step_write += float(row["Counter_Value"])
                        fam_bytes[family(row["Kernel_Name"])] += float(row["Counter_Value"]) * 1024.0
    res = {"_note": __doc__.split("\n\n")[1].replace("\n", " ") if False else
           "rocprofv3 --pmc passes (one counter group per pass) summarised by scripts/pmc_summary.py; per-launch means"}
    for fam in sorted(sums):
        m = {c: sums[fam][c] / counts[fam][c] for c in sums[fam]}
        e = {"launches": max(counts[fam].values()), "per_launch": m}
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            e["hbm_bytes_per_launch"] = (2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0
        if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m and m["TCC_HIT_sum"] + m["TCC_MISS_sum"] > 0:
            e["l2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
        if "SQ_WAVE_CYCLES" in m and m["SQ_WAVE_CYCLES"] > 0:
            for key, c in (("wait_share", "SQ_WAIT_ANY"), ("issue_stall_share", "SQ_WAIT_INST_ANY"),
                           ("active_share", "SQ_ACTIVE_INST_ANY"), ("lds_stall_share", "SQ_WAIT_INST_LDS")):
                if c in m:
                    e[key] = m[c] / m["SQ_WAVE_CYCLES"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and m.get("GRBM_GUI_ACTIVE", 0) > 0:
            e["mfma_busy"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8.0 * 4 * 256)
        if "SQ_LDS_BANK_CONFLICT" in m and m.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
            e["lds_conflict_share"] = m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]
        res[fam] = e
    if step_launches:
        # whole train step: every kernel's 2 * FETCH_SIZE + WRITE_SIZE, divided by the steps seen (one margin3m launch each)
        # the bench line's dominant kernel = every forward / dgrad conv launch (gather + LDS-patch + direct stem forms)
        fd = [k for k in ("conv_gather", "conv_patch", "stem_fwd") if k in res]
        n_fd = sum(res[k]["launches"] for k in fd)
        if n_fd and all("hbm_bytes_per_launch" in res[k] for k in fd):
            res["conv_fwd_dgrad"] = {"launches": n_fd, "families": fd,
                                     "hbm_bytes_per_launch": sum(res[k]["hbm_bytes_per_launch"] * res[k]["launches"] for k in fd) / n_fd}
        res["_step"] = {"train_steps": step_launches,
                        "hbm_bytes_per_step": (2.0 * step_fetch + step_write) * 1024.0 / step_launches,
                        "by_family": {k: v / step_launches for k, v in sorted(fam_bytes.items())}}
    if len(sys.argv) > 1:
        with open(out_path, "w") as f:
            json.dump(res, f, indent=1, sort_keys=True)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "per_launch"} for k, v in res.items() if not k.startswith("_")}, indent=1))


if __name__ == "__main__":
    main()
