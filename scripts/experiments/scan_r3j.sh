OUT=${1:-gpurun_out/scan_r3j}; mkdir -p $OUT
run() { name=$1; shift; env "$@" python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-distmat --no-fp32 2>$OUT/$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],3), 'gather', round(d['roofline']['achieved'],1), 'wgrad', round(d['roofline']['wgrad']['achieved'],1))"; }
for i in 1 2 3; do
run base IEEE_DUMMY=0
run lds48 IEEE_WGRAD_LDS=48
run lds64 IEEE_WGRAD_LDS=64
done
