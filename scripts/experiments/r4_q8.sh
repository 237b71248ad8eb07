cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
O=gpurun_out/r4_q8; mkdir -p $O
for q in 8 4; do
  export GPU_MAX_HW_QUEUES=$q
  rocprofv3 --kernel-trace --output-format csv -d $O/trace$q -- python3 bench.py --steps 8 --warmup 3 --no-roofline-pass --no-cpu-baseline --no-distmat --no-fp32 --no-loader > $O/trace$q.log 2>&1
  grep -o '"dp_path": {[^}]*' $O/trace$q.log | cut -c1-260
  # steps in the trace: 11 main, 25 staged warm-up, 6 unoverlapped, then rounds of 12 plain / 12 staged / 12 unoverlapped: 108 in the rounds
  # from the end: the last round's unoverlapped steps are 1..12, its staged steps 13..24, its plain steps 25..36
  echo "== queues $q: a staged step"; python scripts/phase_table.py $O/trace$q 18 | grep -E "^step|^stream|conv_gather |conv_wgrad |sgd|bn_bwd_apply_totals " 
  echo "== queues $q: a plain step"; python scripts/phase_table.py $O/trace$q 30 | grep -E "^step|^stream"
  python - <<PY
import csv, glob, collections
f = glob.glob("$O/trace$q/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print("queue ids by stream:", sorted({(r["Stream_Id"], r["Queue_Id"]) for r in rows}))
PY
  find $O/trace$q -name "*.csv" -size +1M -delete; find $O/trace$q -name "*.db" -delete
done
