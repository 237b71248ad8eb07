# kernel stats of PLAIN two-stream steps (no event passes in the trace): profiles/rNN_kernel_stats_in_situ.{csv,json}
# usage: bash scripts/in_situ_stats.sh OUTDIR TAG   (GPU box)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp && export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-2}
O=${1:-gpurun_out/in_situ}; TAG=${2:-r04}; mkdir -p $O
STEPS=20; WARM=5
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps $STEPS --warmup $WARM --no-roofline-pass --no-cpu-baseline --no-distmat --no-fp32 --no-dp-path --no-loader --no-config5 > $O/trace.log 2>&1
find $O/trace -name "*_kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_kernel_stats_in_situ.csv
find $O/trace -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $O/kernel_trace_in_situ.csv
python - <<PY
import json, subprocess
line = json.loads([l for l in open("$O/trace.log").read().splitlines() if l.startswith("{")][-1])
B = 64
meta = {"steps_in_trace": $STEPS + $WARM, "command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps $STEPS --warmup $WARM --no-roofline-pass --no-cpu-baseline --no-distmat --no-fp32 --no-dp-path --no-loader --no-config5",
        "build": subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "worktree",
        "ms_per_step_under_the_tracer": line["ms_per_step"],
        # algorithmic conv FLOPs per triple as ieee_net_profile counts them (2*M*N*K of every conv launch): forward 30.762 + dgrad 30.300 (no stem dgrad); weight gradients 30.762
        "fwd_dgrad_flops_per_step": B * 61.061726208e9, "wgrad_flops_per_step": B * 30.762e9}
json.dump(meta, open("$O/${TAG}_kernel_stats_in_situ.json", "w"), indent=1)
print(meta)
PY
