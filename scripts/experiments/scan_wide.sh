mkdir -p gpurun_out/scan5
run() { name=$1; shift; env "$@" IEEE_PROFILE_DUMP=gpurun_out/scan5/$name.csv python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-distmat --no-fp32 2>gpurun_out/scan5/$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['value'],1), round(d['ms_per_step'],3), 'gather TF', round(d['roofline']['achieved'],1))"; }
run base X=1
run w2048 IEEE_GATHER_WIDE=2048
run w1024 IEEE_GATHER_WIDE=1024
run w512 IEEE_GATHER_WIDE=512
run base2 X=1
