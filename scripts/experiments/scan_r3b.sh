OUT=${1:-gpurun_out/scan_r3b}; mkdir -p $OUT
run() { name=$1; shift; env "$@" IEEE_PROFILE_DUMP=$OUT/$name.csv python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-distmat --no-fp32 2>$OUT/$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],3), 'gather', round(d['roofline']['achieved'],1), 'wgrad', round(d['roofline']['wgrad']['achieved'],1))"; }
run base IEEE_GATHER_256=0
run big8 IEEE_GATHER_256=1
run big4 IEEE_GATHER_256_KT=4
run big16 IEEE_GATHER_256_KT=16
run big8_wg256 IEEE_GATHER_256_WG=256
run base_b IEEE_GATHER_256=0
run big8_b IEEE_GATHER_256=1
