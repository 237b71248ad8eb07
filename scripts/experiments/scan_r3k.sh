OUT=${1:-gpurun_out/scan_r3k}; mkdir -p $OUT
run() { name=$1; shift; env "$@" python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-distmat --no-fp32 2>$OUT/$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],3))"; }
for i in 1 2 3 4 5; do
run base IEEE_DUMMY=0
run lds48 IEEE_WGRAD_LDS=48
done
