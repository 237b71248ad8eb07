mkdir -p gpurun_out/scan4
run() { name=$1; shift; env "$@" IEEE_PROFILE_DUMP=gpurun_out/scan4/$name.csv python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-distmat --no-fp32 2>gpurun_out/scan4/$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['value'],1), round(d['ms_per_step'],3), 'gather TF', round(d['roofline']['achieved'],1), 'wgrad TF', round(d['roofline']['wgrad']['achieved'],1))"; }
run base X=1
run s896 IEEE_WGRAD_TARGET_SMALL=896
run s1792 IEEE_WGRAD_TARGET_SMALL=1792
run t896 IEEE_WGRAD_TARGET=896
run base2 X=1
