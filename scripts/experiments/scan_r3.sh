# per-launch conv tables + step time for the round-3 kernel variants: bash scripts/experiments/scan_r3.sh [outdir]
OUT=${1:-gpurun_out/scan_r3}; mkdir -p $OUT
run() { name=$1; shift; env "$@" IEEE_PROFILE_DUMP=$OUT/$name.csv python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-distmat --no-fp32 2>$OUT/$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['ms_per_step'],3), 'gather', round(d['roofline']['achieved'],1), 'wgrad', round(d['roofline']['wgrad']['achieved'],1))"; }
run base IEEE_CONV_PATCH=0
run patch_s1 IEEE_CONV_PATCH=1
run patch_s0 IEEE_PATCH_STYLE=0
run patch_s1_bn128 IEEE_PATCH_BN=128
run patch_s0_bn128 IEEE_PATCH_STYLE=0 IEEE_PATCH_BN=128
run patch_s1_bn64 IEEE_PATCH_BN=64
run pipe5 IEEE_CONV_PATCH=0 IEEE_GATHER_PIPE=5
run pipe5_patch IEEE_GATHER_PIPE=5
run base2 IEEE_CONV_PATCH=0
run patch_s1_b IEEE_CONV_PATCH=1
