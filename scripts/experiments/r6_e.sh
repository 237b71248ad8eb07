# round 6, call e: the range-guard tests, a trace pair for the persistent stem, the ranking kernel's LDS split, stem A/B
mkdir -p gpurun_out/r6_e
timeout 900 python -m pytest tests/test_trajectory_gpu.py -q -k "fixed_point_range" > gpurun_out/r6_e/pytest.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/r6_e/pytest.txt
tail -4 gpurun_out/r6_e/pytest.txt
for d in 0 1 2 3; do echo "== IEEE_RANK_DBG=$d" >> gpurun_out/r6_e/rank.txt; IEEE_RANK_DBG=$d timeout 600 python scripts/rank_probe.py >> gpurun_out/r6_e/rank.txt 2>&1; done
cat gpurun_out/r6_e/rank.txt
timeout 1200 bash scripts/trace_pair.sh gpurun_out/r6_e/stem "IEEE_STEM_WALK=1" "IEEE_STEM_WALK=4" > gpurun_out/r6_e/trace_pair.log 2>&1
grep -E "^==|stem_conv|step span|pack_all|sgd" gpurun_out/r6_e/stem/summary.txt
rm -f gpurun_out/r6_e/stem/trace*.csv
timeout 1500 bash scripts/ab5.sh 6 "IEEE_STEM_WALK=1" "IEEE_STEM_WALK=4" > gpurun_out/r6_e/ab.txt 2>&1
cat gpurun_out/r6_e/ab.txt
