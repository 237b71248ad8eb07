# round-4 evidence run on the GPU box: full GPU tests, per-launch table, PMC passes (train + evaluator), kernel stats of plain
# two-stream steps (the in-situ roofline's cross-check), stream phase table, the default bench line
cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
O=gpurun_out/final_r4; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -n 3 $O/pytest_gpu.log
IEEE_PROFILE_DUMP=$O/launches.csv python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-distmat --no-fp32 --no-loader --no-dp-path > $O/launch_bench.json 2> $O/launch_bench.err
bash scripts/pmc_passes.sh $O/pmc > $O/pmc.log 2>&1
python scripts/pmc_summary.py $O/pmc_summary.json $O/pmc/sq $O/pmc/sq2 $O/pmc/l2 $O/pmc/fetch $O/pmc/write > $O/pmc_summary.txt 2>&1
python scripts/pmc_summary.py --by-grid $O/pmc_by_grid.json $O/pmc/sq $O/pmc/sq2 $O/pmc/l2 $O/pmc/fetch $O/pmc/write > /dev/null 2>&1
python scripts/layer_table.py $O/launches.csv $O/pmc $O/layers.csv 6 > $O/layers.log 2>&1; cat $O/layers.log
bash scripts/in_situ_stats.sh $O/in_situ r04 > $O/in_situ.log 2>&1; tail -n 2 $O/in_situ.log
python scripts/phase_table.py $O/in_situ/trace 8 > $O/phase.txt 2>&1
python scripts/main_gaps.py $O/in_situ/trace 8 > $O/gaps.txt 2>&1
bash scripts/pmc_passes_eval.sh $O/pmc_eval > $O/pmc_eval.log 2>&1
python scripts/pmc_summary.py $O/pmc_eval_summary.json $O/pmc_eval/sq $O/pmc_eval/sq2 $O/pmc_eval/l2 $O/pmc_eval/fetch $O/pmc_eval/write > $O/pmc_eval_summary.txt 2>&1
python scripts/host_enqueue_probe.py > $O/host_enqueue.txt 2>&1; tail -n 1 $O/host_enqueue.txt
mkdir -p profiles; cp $O/in_situ/r04_kernel_stats_in_situ.csv $O/in_situ/r04_kernel_stats_in_situ.json profiles/ 2>/dev/null
python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-400 $O/bench.json
# keep the merge under the 64 MiB cap: the raw per-dispatch counter files are large
find $O -name "*counter_collection.csv" -size +6M -delete; find $O -name "*kernel_trace*.csv" -size +8M -delete; find $O -name "*.db" -delete
du -sh $O
