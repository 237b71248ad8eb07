mkdir -p gpurun_out/scan7
IEEE_PROFILE_DUMP=gpurun_out/scan7/cur.csv python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-distmat --no-fp32 2>/dev/null | tail -1 | cut -c1-100
