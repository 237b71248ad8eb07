OUT=${1:-gpurun_out/scan_r3g}; mkdir -p $OUT
run() { name=$1; shift; env "$@" IEEE_PROFILE_DUMP=$OUT/$name.csv python bench.py --steps 20 --warmup 6 --no-cpu-baseline --no-distmat --no-fp32 2>$OUT/$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],3), 'gather', round(d['roofline']['achieved'],1), 'wgrad', round(d['roofline']['wgrad']['achieved'],1))"; }
for i in 1 2; do
run s1 IEEE_PATCH_STYLE=1
run s0 IEEE_PATCH_STYLE=0
run s0_bn64 IEEE_PATCH_STYLE=0 IEEE_PATCH_BN=64
run s0_bn128 IEEE_PATCH_STYLE=0 IEEE_PATCH_BN=128
done
