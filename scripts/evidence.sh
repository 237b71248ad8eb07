# Evidence of a round on the GPU box, in parts (one gpurun call each; everything lands in gpurun_out/final_<TAG>/):
#   bash scripts/evidence.sh <TAG> tests     the full GPU suite
#   bash scripts/evidence.sh <TAG> pmc       per-launch table + five rocprofv3 --pmc passes of the train step + per-layer table
#   bash scripts/evidence.sh <TAG> trace     rocprofv3 --kernel-trace --stats of plain two-stream steps, stream phase / gap tables,
#                                            host enqueue time, the evaluator's counters
#   bash scripts/evidence.sh <TAG> bench     the default `python bench.py` line
# No trace domain beside --pmc in any counter pass.  scripts/collect_profiles.py <TAG> copies the summaries into profiles/.
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
TAG=${1:-r05}; PART=${2:-bench}
O=gpurun_out/final_$TAG; mkdir -p $O
LEAN="--no-cpu-baseline --no-distmat --no-fp32 --no-loader --no-dp-path --no-config5"
case $PART in
tests)
  timeout 2400 python -m pytest tests -m gpu -q --durations=12 > $O/pytest_gpu.log 2>&1; tail -n 16 $O/pytest_gpu.log | cut -c1-200 ;;
pmc)
  IEEE_PROFILE_DUMP=$O/launches.csv python bench.py --steps 6 --warmup 3 $LEAN > $O/launch_bench.json 2> $O/launch_bench.err
  bash scripts/pmc_passes.sh $O/pmc --steps 3 --warmup 1 --no-roofline-pass $LEAN > $O/pmc.log 2>&1
  python scripts/pmc_summary.py $O/pmc_summary.json $O/pmc/sq $O/pmc/sq2 $O/pmc/l2 $O/pmc/fetch $O/pmc/write > $O/pmc_summary.txt 2>&1
  python scripts/pmc_summary.py --by-grid $O/pmc_by_grid.json $O/pmc/sq $O/pmc/sq2 $O/pmc/l2 $O/pmc/fetch $O/pmc/write > /dev/null 2>&1
  python scripts/layer_table.py $O/launches.csv $O/pmc $O/layers.csv 6 > $O/layers.log 2>&1; tail -n 5 $O/layers.log ;;
trace)
  bash scripts/in_situ_stats.sh $O/in_situ $TAG > $O/in_situ.log 2>&1; tail -n 2 $O/in_situ.log
  python scripts/phase_table.py $O/in_situ/trace 8 > $O/phase.txt 2>&1; head -n 12 $O/phase.txt
  python scripts/main_gaps.py $O/in_situ/trace 8 > $O/gaps.txt 2>&1
  python scripts/host_enqueue_probe.py > $O/host_enqueue.txt 2>&1; tail -n 1 $O/host_enqueue.txt
  bash scripts/pmc_passes_eval.sh $O/pmc_eval > $O/pmc_eval.log 2>&1
  python scripts/pmc_summary.py $O/pmc_eval_summary.json $O/pmc_eval/sq $O/pmc_eval/sq2 $O/pmc_eval/l2 $O/pmc_eval/fetch $O/pmc_eval/write > $O/pmc_eval_summary.txt 2>&1 ;;
bench)
  ( time python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; cat $O/bench.time; cut -c1-300 $O/bench.json; wc -l $O/bench.json ;;
esac
find $O -name "*counter_collection.csv" -size +6M -delete; find $O -name "*kernel_trace*.csv" -size +8M -delete; find $O -name "*.db" -delete
du -sh $O
