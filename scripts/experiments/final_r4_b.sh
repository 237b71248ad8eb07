# round-4 evidence, part B: per-launch table, PMC passes of the train step, per-layer table
cd "$(dirname "$0")/../.." && export TMPDIR=/tmp && export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-2}
O=gpurun_out/final_r4; mkdir -p $O
IEEE_PROFILE_DUMP=$O/launches.csv python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-distmat --no-fp32 --no-loader --no-dp-path > $O/launch_bench.json 2> $O/launch_bench.err
bash scripts/pmc_passes.sh $O/pmc --steps 3 --warmup 1 --no-roofline-pass --no-cpu-baseline --no-distmat --no-fp32 --no-loader --no-dp-path > $O/pmc.log 2>&1
python scripts/pmc_summary.py $O/pmc_summary.json $O/pmc/sq $O/pmc/sq2 $O/pmc/l2 $O/pmc/fetch $O/pmc/write > $O/pmc_summary.txt 2>&1
python scripts/pmc_summary.py --by-grid $O/pmc_by_grid.json $O/pmc/sq $O/pmc/sq2 $O/pmc/l2 $O/pmc/fetch $O/pmc/write > /dev/null 2>&1
python scripts/layer_table.py $O/launches.csv $O/pmc $O/layers.csv 6 > $O/layers.log 2>&1; cat $O/layers.log
find $O -name "*counter_collection.csv" -size +6M -delete; find $O -name "*kernel_trace*.csv" -size +8M -delete; find $O -name "*.db" -delete
du -sh $O
