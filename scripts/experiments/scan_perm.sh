mkdir -p gpurun_out/scan6
for v in 0 1; do IEEE_DGRAD_PERM=$v IEEE_PROFILE_DUMP=gpurun_out/scan6/perm$v.csv python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-distmat --no-fp32 2>/dev/null | tail -1 | cut -c1-120; done
python scripts/variant_compare.py gpurun_out/scan6 dgrad | grep "s2\|shape\|TOTAL" | cut -c1-100
