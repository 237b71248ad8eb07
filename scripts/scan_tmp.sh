for rep in 1 2 3; do
for lib in build/libieee_old.so ieee_amd/libieee_amd.so; do
echo "$lib $(IEEE_AMD_LIB=$PWD/$lib python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-distmat --no-fp32 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["ms_per_step"], d["value"])')"
done
done
