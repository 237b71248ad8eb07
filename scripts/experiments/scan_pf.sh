mkdir -p gpurun_out/scan3
run() { name=$1; shift; env "$@" IEEE_PROFILE_DUMP=gpurun_out/scan3/$name.csv python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-distmat --no-fp32 2>gpurun_out/scan3/$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['value'],1), round(d['ms_per_step'],3), 'gather TF', round(d['roofline']['achieved'],1), 'wgrad TF', round(d['roofline']['wgrad']['achieved'],1), d['config']['loss_last_step'])"; }
run base IEEE_PF=0
run pf1 IEEE_PF=1
run pf2 IEEE_PF=2
run pf3 IEEE_PF=3
run base2 IEEE_PF=0
run pf3b IEEE_PF=3
