# round-4 evidence, part D: the default bench line
cd "$(dirname "$0")/../.." && export TMPDIR=/tmp && export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-2}
O=gpurun_out/final_r4; mkdir -p $O
( time python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; cat $O/bench.time; cut -c1-300 $O/bench.json; wc -l $O/bench.json
