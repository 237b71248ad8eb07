cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
O=gpurun_out/r4_p; mkdir -p $O
for i in 1 2; do
  timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_$i.log 2>&1; echo "run $i rc=$?"; tail -n 1 $O/pytest_$i.log
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
