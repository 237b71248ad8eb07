OUT=${1:-gpurun_out/scan_r3f}; mkdir -p $OUT
run() { name=$1; shift; env "$@" IEEE_PROFILE_DUMP=$OUT/$name.csv python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-distmat --no-fp32 2>$OUT/$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],3), 'gather', round(d['roofline']['achieved'],1), 'wgrad', round(d['roofline']['wgrad']['achieved'],1), 'wgrad_us', round(d['roofline']['wgrad']['avg_launch_us'],1))"; }
for i in 1 2; do
run nopatch IEEE_WGRAD_PATCH=0
run t512 IEEE_WPATCH_TARGET=512
run t768 IEEE_WPATCH_TARGET=768
run t1024 IEEE_WPATCH_TARGET=1024
run t1536 IEEE_WPATCH_TARGET=1536
done
