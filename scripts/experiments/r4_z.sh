cd "$(dirname "$0")/../.." && export TMPDIR=/tmp
O=gpurun_out/r4_z; mkdir -p $O
timeout 900 python -m pytest tests/test_dp_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; tail -n 4 $O/pytest.log | cut -c1-300
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-distmat --no-fp32 --no-loader --no-roofline-pass 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], json.dumps(d['dp_path'])[:700])"
# two ranks on one GPU (gloo): the calibration and the N > 1 line
IEEE_DIST_BACKEND=gloo IEEE_FORCE_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline --no-distmat --no-fp32 --no-loader --no-roofline-pass > $O/n2.json 2> $O/n2.err; tail -c 1500 $O/n2.json; tail -n 3 $O/n2.err | cut -c1-300
