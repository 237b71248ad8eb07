OUT=${1:-gpurun_out/scan_r3d}; mkdir -p $OUT
run() { name=$1; shift; env "$@" python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-distmat --no-fp32 2>$OUT/$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],3), 'gather', round(d['roofline']['achieved'],1), 'wgrad', round(d['roofline']['wgrad']['achieved'],1), 'wgrad_us', round(d['roofline']['wgrad']['avg_launch_us'],1))"; }
for i in 1 2 3; do
run nostem IEEE_STEM_DIRECT=0
run stem IEEE_STEM_DIRECT=1
done
