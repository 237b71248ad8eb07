# round 6, call c: the whole GPU suite on the defaults (explicit-argument conv calls, range-guard degrade, loader end-of-pass
# marker, paired head GEMMs; both in-launch weight-gradient reductions off), the two optional wgrad forms through the executor,
# and the head-pair A/B
mkdir -p gpurun_out/r6_c
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r6_c/pytest.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/r6_c/pytest.txt
tail -5 gpurun_out/r6_c/pytest.txt
for v in IEEE_WGRAD_CHAIN=1 IEEE_WGRAD_FOLD=1 IEEE_HEAD_PAIRS=0; do
  env $v timeout 900 python -m pytest tests/test_model_gpu.py tests/test_backward_units_gpu.py -x -q -k "not profile" > gpurun_out/r6_c/pytest_$v.txt 2>&1
  echo "$v rc=$?" >> gpurun_out/r6_c/pytest.txt
  tail -2 gpurun_out/r6_c/pytest_$v.txt
done
timeout 1500 bash scripts/ab5.sh 4 "IEEE_HEAD_PAIRS=0" "IEEE_HEAD_PAIRS=1" > gpurun_out/r6_c/ab.txt 2>&1
cat gpurun_out/r6_c/ab.txt
