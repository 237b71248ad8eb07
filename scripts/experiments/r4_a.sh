# round 4, call A: fixed-channel BatchNorm kernels (tests, isolated bandwidth, step A/B) + the training-trajectory probe
cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
O=gpurun_out/r4_a; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py tests/test_conv_gpu.py -m gpu -q -x > $O/pytest_kernels.log 2>&1; tail -n 3 $O/pytest_kernels.log
IEEE_BN_FIXED=0 python scripts/bn_probe.py > $O/bn_probe_plain.txt 2>&1
python scripts/bn_probe.py > $O/bn_probe_fixed.txt 2>&1
IEEE_BN_NT=1 python scripts/bn_probe.py > $O/bn_probe_fixed_nt.txt 2>&1
paste -d'\n' $O/bn_probe_plain.txt $O/bn_probe_fixed.txt $O/bn_probe_fixed_nt.txt
bash scripts/ab_env.sh "IEEE_BN_FIXED=0" "IEEE_BN_FIXED=1" 3 2>&1 | tee $O/ab_fixed.txt
bash scripts/ab_env.sh "IEEE_BN_NT=0" "IEEE_BN_NT=1" 2 2>&1 | tee $O/ab_nt.txt
python scripts/trajectory_probe.py > $O/trajectory.txt 2>&1; tail -n 40 $O/trajectory.txt
