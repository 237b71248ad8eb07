#!/bin/bash
# per-layer table of the forward + dgrad launches under different 128x64-tile thresholds (IEEE_GATHER_NARROW_WG) and tile groups
O=gpurun_out/scan_m; rm -rf $O; mkdir -p $O
run() { name=$1; shift; env "$@" IEEE_PROFILE_DUMP=$O/$name.csv python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-distmat --no-fp32 > /dev/null 2>&1; }
run base X=1
run nw0 IEEE_GATHER_NARROW_WG=0
run nw1024 IEEE_GATHER_NARROW_WG=1024
run nw2048 IEEE_GATHER_NARROW_WG=2048
run nwall IEEE_GATHER_NARROW_WG=100000000
run grp4 IEEE_TILE_GROUP=4
run grp16 IEEE_TILE_GROUP=16
python scripts/variant_compare.py $O fwd,dgrad | tee $O/compare.txt
