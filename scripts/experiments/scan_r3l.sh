#!/bin/bash
# per-layer table of the forward + dgrad launches under deeper LDS-DMA rings (all launches), against the plan's choice
O=gpurun_out/scan_l; rm -rf $O; mkdir -p $O
run() { name=$1; shift; env "$@" IEEE_PROFILE_DUMP=$O/$name.csv python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-distmat --no-fp32 > /dev/null 2>&1; }
run base X=1
run pipe2 IEEE_GATHER_PIPE=2
run pipe3 IEEE_GATHER_PIPE=3
run pipe0 IEEE_GATHER_PIPE=0
python scripts/variant_compare.py $O fwd,dgrad | tee $O/compare.txt
