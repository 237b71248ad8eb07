# round 4, call D: trajectory + fixbase tests, eval-noise scan, in-situ profile mode sanity
cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
O=gpurun_out/r4_d; mkdir -p $O
python -m pytest tests/test_trajectory_gpu.py tests/test_fixbase_gpu.py -m gpu -q -s > $O/pytest_new.log 2>&1; grep -n "passed\|failed\|Error\|error\|tail on the native\|20 steps\|frozen head\|max |smoothed\|loss .* ->" $O/pytest_new.log | cut -c1-1500
python scripts/trajectory_probe.py > $O/trajectory.txt 2>&1; head -n 8 $O/trajectory.txt
