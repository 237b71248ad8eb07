# round 4, call C: the new trajectory tests, then the whole GPU suite, then the evaluation-noise scan of the probe
cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
O=gpurun_out/r4_c; mkdir -p $O
python -m pytest tests/test_trajectory_gpu.py -m gpu -q -x -s > $O/pytest_traj.log 2>&1; tail -n 25 $O/pytest_traj.log
python -m pytest tests -m gpu -q -x --deselect tests/test_trajectory_gpu.py > $O/pytest_gpu.log 2>&1; tail -n 5 $O/pytest_gpu.log
EPOCHS=12 python scripts/trajectory_probe.py > $O/trajectory.txt 2>&1; head -n 8 $O/trajectory.txt
