# round 6, call i: phase stagger and dual-issue variants of conv_gather: serialized per-launch tables, then the step
mkdir -p gpurun_out/r6_i
LEAN="--no-cpu-baseline --no-distmat --no-fp32 --no-loader --no-dp-path --no-config5"
run() { env $2 IEEE_PROFILE_DUMP=gpurun_out/r6_i/$1.csv timeout 600 python bench.py --steps 6 --warmup 3 $LEAN > gpurun_out/r6_i/$1.json 2> gpurun_out/r6_i/$1.err; }
run base IEEE_NOP=1
run stag2 IEEE_GATHER_STAGGER=2
run stag4 IEEE_GATHER_STAGGER=4
run stag8 IEEE_GATHER_STAGGER=8
run stag16 IEEE_GATHER_STAGGER=16
run dual8 IEEE_GATHER_DUAL=8
run dual4 IEEE_GATHER_DUAL=4
python scripts/experiments/r6_variants_compare.py base=gpurun_out/r6_i/base.csv s2=gpurun_out/r6_i/stag2.csv s4=gpurun_out/r6_i/stag4.csv s8=gpurun_out/r6_i/stag8.csv s16=gpurun_out/r6_i/stag16.csv dual8=gpurun_out/r6_i/dual8.csv dual4=gpurun_out/r6_i/dual4.csv > gpurun_out/r6_i/compare.txt 2>&1
cat gpurun_out/r6_i/compare.txt
timeout 1500 bash scripts/ab5.sh 3 "IEEE_NOP=1" "IEEE_GATHER_STAGGER=4" "IEEE_GATHER_STAGGER=8" "IEEE_GATHER_DUAL=8" > gpurun_out/r6_i/ab.txt 2>&1
cat gpurun_out/r6_i/ab.txt
