# per-launch conv timing tables of the full training step under different kernel variants (tuning aid)
# usage (on the GPU box): bash scripts/experiments/variant_scan.sh name "ENV=.. ENV=.." ...
mkdir -p gpurun_out/scan
while [ $# -gt 1 ]; do
  name=$1; envs=$2; shift 2
  env $envs IEEE_PROFILE_DUMP=gpurun_out/scan/$name.csv python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-distmat 2>gpurun_out/scan/$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['value'],1), round(d['ms_per_step'],3))"
done
