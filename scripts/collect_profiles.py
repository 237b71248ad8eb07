"""Copy the summaries of an evidence run (scripts/evidence.sh <TAG> ...: gpurun_out/final_<TAG>/) into profiles/ under the
round's names, stamping the build they were taken on (the GPU box has no .git: the stamp is this checkout's HEAD, plus "+dirty"
when the tree differs from it).   python scripts/collect_profiles.py r05"""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
src = os.path.join(ROOT, "gpurun_out", "final_" + tag)
dst = os.path.join(ROOT, "profiles")
head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip()
dirty = subprocess.run(["git", "status", "--porcelain", "--", "ieee_amd", "include", "bench.py"], capture_output=True, text=True,
                       cwd=ROOT).stdout.strip()
build = head + ("+dirty" if dirty else "")
plain = [("bench.json", "%s_bench_line.json"), ("in_situ/%s_kernel_stats_in_situ.csv" % tag, "%s_kernel_stats_in_situ.csv"),
         ("pmc/kernel_stats.csv", "%s_kernel_stats.csv"), ("layers.csv", "%s_layers.csv"), ("pmc_by_grid.json", "%s_pmc_by_grid.json"),
         ("pmc_eval/kernel_stats.csv", "%s_evaluator_kernel_stats.csv"), ("phase.txt", "%s_stream_phase.txt"),
         ("gaps.txt", "%s_stream_gaps.txt"), ("host_enqueue.txt", "%s_host_enqueue.txt")]
stamped = [("in_situ/%s_kernel_stats_in_situ.json" % tag, "%s_kernel_stats_in_situ.json", "build"),
           ("pmc_summary.json", "%s_pmc_summary.json", "_build"), ("pmc_eval_summary.json", "%s_pmc_evaluator.json", "_build")]
for a, b in plain:
    p = os.path.join(src, a)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, b % tag))
        print("copied", a, "->", b % tag)
    else:
        print("missing", a)
for a, b, key in stamped:
    p = os.path.join(src, a)
    if os.path.exists(p):
        d = json.load(open(p))
        d[key] = build
        json.dump(d, open(os.path.join(dst, b % tag), "w"), indent=1)
        print("copied", a, "->", b % tag, "(build %s)" % build)
    else:
        print("missing", a)
