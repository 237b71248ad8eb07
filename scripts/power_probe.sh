# clocks / power of the GPU while the train step runs (is the step power- or clock-limited?): bash scripts/power_probe.sh OUT
cd "$(dirname "$0")/.." && O=${1:-gpurun_out/power.txt}
python bench.py --steps 600 --warmup 10 --no-cpu-baseline --no-distmat --no-fp32 --no-loader --no-dp-path --no-config5 --no-roofline-pass > $O.bench 2>/dev/null &
BP=$!
sleep 12
for i in $(seq 1 8); do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -i -E "sclk|mclk|fclk|power|junction|edge" >> $O; echo "--" >> $O; sleep 0.7; done
wait $BP
echo "idle:" >> $O; sleep 3; rocm-smi --showclocks --showpower 2>/dev/null | grep -i -E "sclk|mclk|power" >> $O
rocm-smi --showmaxpower --showclkfrq 2>/dev/null | head -60 >> $O
cat $O.bench | cut -c1-200 >> $O
