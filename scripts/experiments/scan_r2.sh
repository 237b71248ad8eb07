# round-2 variant scan (GPU box): per-launch conv tables of the full step under pipeline / tile overrides
mkdir -p gpurun_out/scan2
run() { name=$1; shift; env "$@" IEEE_PROFILE_DUMP=gpurun_out/scan2/$name.csv python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-distmat --no-fp32 2>gpurun_out/scan2/$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['value'],1), round(d['ms_per_step'],3), 'gather TF', round(d['roofline']['achieved'],1), 'wgrad TF', round(d['roofline']['wgrad']['achieved'],1))"; }
run base X=1
run p2_800 IEEE_GATHER_PIPE=2 IEEE_GATHER_MAXWG=800
run p2_1600 IEEE_GATHER_PIPE=2 IEEE_GATHER_MAXWG=1600
run p3_800 IEEE_GATHER_PIPE=3 IEEE_GATHER_MAXWG=800
run p3_1600 IEEE_GATHER_PIPE=3 IEEE_GATHER_MAXWG=1600
run p5_1600 IEEE_GATHER_PIPE=5 IEEE_GATHER_MAXWG=1600
run wp2 IEEE_WGRAD_PIPE=2
run wp3 IEEE_WGRAD_PIPE=3
run wt640 IEEE_WGRAD_TARGET=640
run wt320 IEEE_WGRAD_TARGET=320
run base2 X=1
