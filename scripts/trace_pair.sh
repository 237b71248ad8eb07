# two kernel traces of plain steps on one box, one per environment, with their phase tables:
#   bash scripts/trace_pair.sh OUTDIR "ENV_A=.." "ENV_B=.."
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
O=$1; shift; mkdir -p $O
i=0
for e in "$@"; do
  i=$((i+1))
  for kv in $e; do export "$kv"; done
  rocprofv3 --kernel-trace --output-format csv -d $O/t$i -- python3 bench.py --steps 12 --warmup 5 --no-roofline-pass --no-cpu-baseline --no-distmat --no-fp32 --no-dp-path --no-loader --no-config5 > $O/t$i.log 2>&1
  for kv in $e; do unset "${kv%%=*}"; done
  python scripts/phase_table.py $O/t$i > $O/phase$i.txt 2>&1
  find $O/t$i -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $O/trace$i.csv
  rm -rf $O/t$i
  echo "== $e" >> $O/summary.txt; head -30 $O/phase$i.txt >> $O/summary.txt
done
