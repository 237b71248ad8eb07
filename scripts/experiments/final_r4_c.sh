# round-4 evidence, part C: kernel stats of plain two-stream steps, stream phase table, evaluator counters, host enqueue time
cd "$(dirname "$0")/../.." && export TMPDIR=/tmp && export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-2}
O=gpurun_out/final_r4; mkdir -p $O
bash scripts/in_situ_stats.sh $O/in_situ r04 > $O/in_situ.log 2>&1; tail -n 2 $O/in_situ.log
python scripts/phase_table.py $O/in_situ/trace 8 > $O/phase.txt 2>&1; head -n 12 $O/phase.txt
python scripts/main_gaps.py $O/in_situ/trace 8 > $O/gaps.txt 2>&1
python scripts/host_enqueue_probe.py > $O/host_enqueue.txt 2>&1; tail -n 1 $O/host_enqueue.txt
bash scripts/pmc_passes_eval.sh $O/pmc_eval > $O/pmc_eval.log 2>&1
python scripts/pmc_summary.py $O/pmc_eval_summary.json $O/pmc_eval/sq $O/pmc_eval/sq2 $O/pmc_eval/l2 $O/pmc_eval/fetch $O/pmc_eval/write > $O/pmc_eval_summary.txt 2>&1
find $O -name "*counter_collection.csv" -size +6M -delete; find $O -name "*kernel_trace*.csv" -size +8M -delete; find $O -name "*.db" -delete
du -sh $O
