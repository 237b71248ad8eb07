# round 4, call B: which of the fixed-channel BatchNorm passes pay (isolated + in the step)
cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
O=gpurun_out/r4_b; mkdir -p $O
for v in "IEEE_BN_FIXED=0" "IEEE_BN_FIXED=15 IEEE_BN_UNROLL=1" "IEEE_BN_FIXED=15 IEEE_BN_UNROLL=2"; do
  echo "== $v"; env $v python scripts/bn_probe.py 2>/dev/null
done | tee $O/bn_probe.txt
run() { env $1 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-distmat --no-fp32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['ms_per_step'],3))"; }
for i in 1 2 3; do
  for v in "IEEE_BN_FIXED=0" "IEEE_BN_FIXED=8 IEEE_BN_UNROLL=1" "IEEE_BN_FIXED=8 IEEE_BN_UNROLL=2" "IEEE_BN_FIXED=10 IEEE_BN_UNROLL=1" "IEEE_BN_FIXED=14 IEEE_BN_UNROLL=1" "IEEE_BN_FIXED=15 IEEE_BN_UNROLL=1" "IEEE_BN_FIXED=1 IEEE_BN_UNROLL=1"; do run "$v"; done
done | tee $O/ab.txt
python scripts/trajectory_probe.py > $O/trajectory.txt 2>&1; head -n 12 $O/trajectory.txt
