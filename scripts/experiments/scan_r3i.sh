OUT=${1:-gpurun_out/scan_r3i}; mkdir -p $OUT
run() { name=$1; shift; env "$@" python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-distmat --no-fp32 2>$OUT/$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],3), 'gather', round(d['roofline']['achieved'],1), 'wgrad', round(d['roofline']['wgrad']['achieved'],1))"; }
for i in 1 2; do
run base IEEE_DUMMY=0
run sgd8k IEEE_HEAD_EW_BLOCKS=8192
run sgd16k IEEE_HEAD_EW_BLOCKS=16384
run wt320 IEEE_WGRAD_TARGET=320
run wt640 IEEE_WGRAD_TARGET=640
run wp320 IEEE_WPATCH_TARGET=320
run narrow768 IEEE_GATHER_NARROW_WG=768
run narrow384 IEEE_GATHER_NARROW_WG=384
run tg4 IEEE_TILE_GROUP=4
run tg16 IEEE_TILE_GROUP=16
done
