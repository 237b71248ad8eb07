# round-3 evidence run on the GPU box: full GPU tests, PMC passes (train + evaluator), per-launch table, kernel stats, bench line
cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
O=gpurun_out/final_r3; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -n 3 $O/pytest_gpu.log
IEEE_PROFILE_DUMP=$O/launches.csv python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-distmat --no-fp32 > $O/launch_bench.json 2> $O/launch_bench.err
bash scripts/pmc_passes.sh $O/pmc > $O/pmc.log 2>&1
python scripts/pmc_summary.py $O/pmc_summary.json $O/pmc/sq $O/pmc/sq2 $O/pmc/l2 $O/pmc/fetch $O/pmc/write > $O/pmc_summary.txt 2>&1
python scripts/pmc_summary.py --by-grid $O/pmc_by_grid.json $O/pmc/sq $O/pmc/sq2 $O/pmc/l2 $O/pmc/fetch $O/pmc/write > /dev/null 2>&1
python scripts/layer_table.py $O/launches.csv $O/pmc $O/layers.csv 6 > $O/layers.log 2>&1; cat $O/layers.log
bash scripts/pmc_passes_eval.sh $O/pmc_eval > $O/pmc_eval.log 2>&1
python scripts/pmc_summary.py $O/pmc_eval_summary.json $O/pmc_eval/sq $O/pmc_eval/sq2 $O/pmc_eval/l2 $O/pmc_eval/fetch $O/pmc_eval/write > $O/pmc_eval_summary.txt 2>&1
python scripts/phase_table.py $O/pmc/trace 14 > $O/phase.txt 2>&1
python scripts/main_gaps.py $O/pmc/trace 14 > $O/gaps.txt 2>&1
python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
# keep the merge under the 64 MiB cap: the raw per-dispatch counter files are large
find $O -name "*counter_collection.csv" -size +6M -delete; find $O -name "*kernel_trace.csv" -size +8M -delete; find $O -name "*.db" -delete
du -sh $O
