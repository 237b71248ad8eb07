cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
O=gpurun_out/r4_u; mkdir -p $O
bash scripts/in_situ_stats.sh $O/in_situ r04 > $O/in_situ.log 2>&1; tail -n 2 $O/in_situ.log
python scripts/phase_table.py $O/in_situ/trace 8 > $O/phase.txt 2>&1; head -n 60 $O/phase.txt
python scripts/main_gaps.py $O/in_situ/trace 8 > $O/gaps.txt 2>&1
find $O -name "*kernel_trace*.csv" -size +8M -delete; find $O -name "*.db" -delete
du -sh $O
