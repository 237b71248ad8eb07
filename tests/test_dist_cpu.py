"""CPU, world_size 2 over gloo: the N>1 data path (ieee_amd/dist.py) — identity-aligned sharding, the
single flat all-reduce, and the loss scaling that makes the sum of rank-local gradients equal the
gradient of the reference's global loss (CE = batch mean, 3M = sum over identities)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ieee_amd import dist as ddp


def test_shard_bounds_cover_batch_on_identity_boundaries():
    for B, K, W in ((512, 4, 8), (64, 4, 2), (24, 4, 4), (8, 4, 2), (40, 4, 3)):
        spans = [ddp.shard_bounds(B, K, W, r) for r in range(W)]
        assert spans[0][0] == 0 and spans[-1][1] == B
        for (a, b), (c, d) in zip(spans, spans[1:]):
            assert b == c
        assert all((b - a) % K == 0 and a % K == 0 for a, b in spans)
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= K
    with pytest.raises(AssertionError):
        ddp.shard_bounds(10, 4, 2, 0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret, B=16, C=11, D=32):
    from oracle import model as om
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank))
    w, r, _ = ddp.init_from_env(backend="gloo")
    assert (w, r) == (world, rank) and ddp.world_size() == world and ddp.rank() == rank
    torch.manual_seed(0)                      # same global batch on every rank
    K = 4
    logits = torch.randn(6, B, C)
    feats = torch.nn.functional.normalize(torch.randn(3, B, D), dim=2)
    pids = torch.arange(B) // K
    data = {"img": [torch.arange(B).float().view(B, 1, 1, 1).expand(B, 3, 2, 2)] * 3, "pid": pids,
            "camid": torch.zeros(B), "timeid": torch.zeros(B), "impath": "x"}
    shard = ddp.shard_batch(data, K)
    a, b = ddp.shard_bounds(B, K, world, rank)
    assert torch.equal(shard["pid"], pids[a:b]) and shard["img"][0].shape[0] == b - a and shard["impath"] == "x"
    assert float(shard["img"][1][0, 0, 0, 0]) == a
    # global loss of the reference: sum_h CE_h (batch mean) + 3M (sum over identities)
    lg = logits.clone().requires_grad_(True)
    fg = feats.clone().requires_grad_(True)
    loss = sum(om.cross_entropy_ls(lg[h], pids, C) for h in range(6)) + om.margin3m(fg[0], fg[1], fg[2], pids, 1.0)
    loss.backward()
    # rank-local loss with the DP scaling rule; gradients land in a flat buffer that is all-reduced once
    ll = logits[:, a:b].clone().requires_grad_(True)
    fl = feats[:, a:b].clone().requires_grad_(True)
    lp = pids[a:b]
    # CE is a batch MEAN: each rank weighs its mean by B_local / B_global (uneven shards when identities % world != 0)
    assert abs(ddp.ce_grad_scale(b - a) - (b - a) / B) < 1e-12 and ddp.global_rows(b - a) == B
    local = ddp.ce_grad_scale(b - a, B) * sum(om.cross_entropy_ls(ll[h], lp, C) for h in range(6)) \
        + om.margin3m(fl[0], fl[1], fl[2], lp, 1.0)
    local.backward()
    flat = torch.zeros(6 * B * C + 3 * B * D)
    gl = torch.zeros(6, B, C)
    gf = torch.zeros(3, B, D)
    gl[:, a:b] = ll.grad
    gf[:, a:b] = fl.grad
    flat[:6 * B * C] = gl.flatten()
    flat[6 * B * C:] = gf.flatten()
    ddp.allreduce_sum_(flat)
    torch.testing.assert_close(flat[:6 * B * C].view(6, B, C), lg.grad, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(flat[6 * B * C:].view(3, B, D), fg.grad, rtol=1e-5, atol=1e-7)
    vec = torch.ones(9)
    ddp.reduce_summary_(vec)
    assert float(vec[0]) == world
    dist.destroy_process_group()
    ret[rank] = 1


# (3, 32): 8 identities over 3 ranks = shards of 12, 12, 8 rows.  (4, 128, 750, 768): BASELINE config 5's shape -- 4 ranks x 32
# triples, the 750 identities of Market1501-multimodal, the 768-wide per-modality descriptors the 3M loss sees
# (8, 512, 171, 768): BASELINE config 3 -- 8 ranks x 64 triples, the 171 identities of RGBNT201
@pytest.mark.parametrize("world,B,C,D", [(2, 16, 11, 32), (3, 32, 11, 32), (4, 128, 750, 768), (8, 512, 171, 768)])
def test_gloo_allreduce_and_loss_scaling(world, B, C, D):
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), ret, B, C, D), nprocs=world, join=True)
        assert dict(ret) == {r: 1 for r in range(world)}


# ---- replica synchronisation and the sharded feature extraction of the evaluator
class _FlatModel(object):
    """the three flat buffers dist.sync_replicas works on (IEEE3modalPart keeps exactly these)"""

    def __init__(self, seed):
        g = torch.Generator().manual_seed(seed)
        self._flat_params = torch.randn(1000, generator=g)
        self._flat_buffers = torch.randn(50, generator=g)
        self._flat_counters = torch.randint(0, 9, (7,), generator=g)
        self.invalidated = 0

    def invalidate_eval_cache(self):
        self.invalidated += 1


class _FlatOpt(object):
    def __init__(self, seed):
        self.buf = torch.full((1000,), float(seed))

    def flat_state(self):
        return [self.buf]


def _sync_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank))
    ddp.init_from_env(backend="gloo")
    ref = _FlatModel(100)                                  # what rank 0 holds
    m, opt = _FlatModel(100 + rank), _FlatOpt(rank)        # every rank starts from its own draw
    own_params = m._flat_params.clone()
    ddp.sync_replicas(m, buffers_only=True)                # evaluation: rank 0's running statistics, parameters untouched
    assert torch.equal(m._flat_buffers, ref._flat_buffers) and torch.equal(m._flat_counters, ref._flat_counters)
    assert torch.equal(m._flat_params, own_params) and m.invalidated == 1
    ddp.sync_replicas(m, opt)                              # first train step: everything, optimizer state included
    assert torch.equal(m._flat_params, ref._flat_params) and torch.equal(opt.buf, torch.zeros(1000))
    # any other nn.Module with a torch.optim optimizer (the generic autograd path): parameters, buffers and the
    # optimizer's state tensors are broadcast one by one
    torch.manual_seed(50 + rank)
    net = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.BatchNorm1d(3))
    sgd = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9)
    net(torch.randn(5, 4)).sum().backward()
    sgd.step()                                             # momentum buffers exist now, different on every rank
    ddp.sync_replicas(net, sgd)
    mine = [t.clone() for t in net.state_dict().values()] + [sgd.state[p]["momentum_buffer"].clone() for p in net.parameters()]
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    for other in everyone:
        assert all(torch.equal(a, b) for a, b in zip(other, everyone[0]))
    # the global batch size is asked with a collective that every rank enters on every call (no rank-local cache):
    # rank 0 repeats its size while the other ranks change theirs
    for step in range(3):
        local = 8 if rank == 0 else 4 + step
        assert ddp.global_rows(local) == 8 + (world - 1) * (4 + step)
    # sharded feature extraction: 5 loader batches of uneven size, rank r holds batches r, r + world, ...
    rows = [4, 4, 3, 4, 1]
    full = torch.arange(sum(rows) * 6, dtype=torch.float32).view(-1, 6)
    starts = [sum(rows[:b]) for b in range(len(rows))]
    local = {b: full[starts[b]:starts[b] + rows[b]].clone() for b in range(rank, len(rows), world)}
    out = ddp.gather_feature_batches(local, rows, 6, torch.device("cpu"))
    assert torch.equal(out, full)
    # ... and a gallery-sized walk: 157 ragged batches (the index-tensor plan, no per-batch copies)
    g = torch.Generator().manual_seed(5)
    rows = torch.randint(1, 65, (157,), generator=g).tolist()
    full = torch.arange(sum(rows) * 3, dtype=torch.float32).view(-1, 3)
    starts = [sum(rows[:b]) for b in range(len(rows))]
    local = {b: full[starts[b]:starts[b] + rows[b]].clone() for b in range(rank, len(rows), world)}
    assert torch.equal(ddp.gather_feature_batches(local, rows, 3, torch.device("cpu")), full)
    dist.destroy_process_group()
    ret[rank] = 1


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_replica_sync_and_sharded_feature_gather(world):
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_sync_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        assert dict(ret) == {r: 1 for r in range(world)}


# ---- query-sharded evaluator: host logic + the single 22-number all-reduce
def _counts_np(distmat, q_pids, g_pids, q_camids, g_camids, max_rank):
    """per-shard checker built on the oracle: sums of the per-query CMC rows / AP over the valid queries"""
    import numpy as np
    from oracle import evaluator as oe
    counts, valid, ap_sum = np.zeros(max_rank, dtype=np.int64), 0, 0.0
    for i in range(distmat.shape[0]):
        try:
            cmc, ap = oe.rank_market1501_np(distmat[i:i + 1], q_pids[i:i + 1], g_pids, q_camids[i:i + 1], g_camids, max_rank)
        except AssertionError:      # this query's identity does not appear in the gallery
            continue
        counts += np.rint(cmc).astype(np.int64)
        valid += 1
        ap_sum += ap
    return counts, float(valid), ap_sum


def _eval_inputs():
    import numpy as np
    rng = np.random.RandomState(3)
    Q, G, D = 37, 211, 16
    qf = rng.randint(-3, 4, size=(Q, D)).astype(np.float32) + rng.rand(Q, D).astype(np.float32) * 1e-3
    gf = rng.randint(-3, 4, size=(G, D)).astype(np.float32) + rng.rand(G, D).astype(np.float32) * 1e-3
    q_pids, g_pids = rng.randint(0, 12, Q), rng.randint(0, 10, G)     # pids 10, 11 never appear in the gallery
    q_cam, g_cam = rng.randint(0, 4, Q), rng.randint(0, 4, G)
    return qf, gf, q_pids, g_pids, q_cam, g_cam


def _eval_worker(rank, world, port, ret):
    import numpy as np
    from oracle import evaluator as oe
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank))
    ddp.init_from_env(backend="gloo")
    qf, gf, qp, gp, qc, gc = _eval_inputs()
    cmc, m_ap = ddp.sharded_evaluate_rank(qf, gf, qp, gp, qc, gc, max_rank=20, distmat_fn=oe.sqeuclid_np,
                                          counts_fn=_counts_np)
    ref_cmc, ref_map = oe.rank_market1501_np(oe.sqeuclid_np(qf, gf), qp, gp, qc, gc, 20)
    np.testing.assert_allclose(cmc, ref_cmc, rtol=0, atol=1e-7)
    assert abs(m_ap - ref_map) < 1e-12
    dist.destroy_process_group()
    ret[rank] = (cmc.tolist(), m_ap)


def test_query_shards_cover_all_rows():
    for Q, W in ((10000, 8), (37, 2), (3, 4), (0, 2)):
        spans = [ddp.query_shard(Q, W, r) for r in range(W)]
        assert spans[0][0] == 0 and spans[-1][1] == Q
        assert all(b == c for (_, b), (c, _) in zip(spans, spans[1:]))
        assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


@pytest.mark.parametrize("world", [2, 3, 8])
def test_query_sharded_evaluator_matches_single_process(world):
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_eval_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        out = dict(ret)
        assert len(out) == world and all(out[r] == out[0] for r in out)      # every rank reports the same result


# ---- launching N ranks from a plain `python` command (ieee_amd.dist.launch; what `python bench.py --gpus N` does) ----------
_RANK_SCRIPT = r"""
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from ieee_amd import dist as ddp
world, rank, local = ddp.init_from_env(backend="gloo")
assert os.environ["GPU_MAX_HW_QUEUES"] == %r, os.environ["GPU_MAX_HW_QUEUES"]
if rank == int(os.environ.get("FAIL_RANK", "-1")):
    sys.exit(7)
t = torch.tensor([float(rank + 1)])
dist.all_reduce(t)
print("rank %%d of %%d: sum %%d" %% (rank, world, int(t.item())), flush=True)
dist.destroy_process_group()
"""


def _launch_parent(tmp_path, world, extra_env=None, queues="1", call_queues="None"):
    """a fresh interpreter plays the launching process (launch() refuses a parent that holds a HIP context, and rank 0
    inherits the launcher's stdout -- here a pipe)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT % (root, queues))
    parent = ("import sys; sys.path.insert(0, %r); from ieee_amd import dist as ddp; "
              "sys.exit(ddp.launch([sys.executable, %r], %d, queues=%s, grace=3.0))" % (root, str(script), world, call_queues))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "GPU_MAX_HW_QUEUES"):
        env.pop(k, None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, "-c", parent], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)


def test_launch_argv_starts_one_rank_per_process_and_relays_rank0(tmp_path):
    r = _launch_parent(tmp_path, 4)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    # ONE line on the launcher's stdout (rank 0's); the other ranks' stdout went to stderr
    # (gloo announces its connections on stdout; bench.py points descriptor 1 at stderr until its JSON line for that reason)
    assert [l for l in r.stdout.decode().strip().splitlines() if not l.startswith("[Gloo]")] == ["rank 0 of 4: sum 10"]
    err = r.stderr.decode()
    assert all(("rank %d of 4: sum 10" % k) in err for k in (1, 2, 3))


def test_launch_eight_ranks_like_one_node(tmp_path):
    """BASELINE config 3's rank count through the same path `python bench.py --gpus 8` takes"""
    r = _launch_parent(tmp_path, 8)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert [l for l in r.stdout.decode().strip().splitlines() if not l.startswith("[Gloo]")] == ["rank 0 of 8: sum 36"]


def test_launch_keeps_an_exported_queue_count(tmp_path):
    r = _launch_parent(tmp_path, 2, queues="3", call_queues="'3'")
    assert r.returncode == 0, r.stderr.decode()[-2000:]


def test_launch_reports_the_failing_rank_and_ends_the_others(tmp_path):
    import time
    t0 = time.time()
    r = _launch_parent(tmp_path, 3, extra_env={"FAIL_RANK": "1"})
    assert r.returncode == 7, (r.returncode, r.stderr.decode()[-2000:])
    assert time.time() - t0 < 120              # the surviving ranks sat in a collective with a dead peer: ended after `grace`


def _launch_fn(rank, world, out_dir):
    w, r, _ = ddp.init_from_env(backend="gloo")
    assert (w, r) == (world, rank) and os.environ["GPU_MAX_HW_QUEUES"] == "1"
    t = torch.tensor([float(rank)])
    dist.all_reduce(t)
    open(os.path.join(out_dir, "rank%d" % rank), "w").write(str(int(t.item())))
    dist.destroy_process_group()


def test_launch_callable_spawns_fresh_interpreters(tmp_path):
    before = dict(os.environ)
    assert ddp.launch(_launch_fn, 3, args=(str(tmp_path),)) == 0
    assert dict(os.environ) == before          # the launcher's own environment is untouched
    assert sorted(os.listdir(str(tmp_path))) == ["rank0", "rank1", "rank2"]
    assert all(open(os.path.join(str(tmp_path), "rank%d" % r)).read() == "3" for r in range(3))


def test_bench_self_launch_fails_loudly_without_a_gpu():
    """`python bench.py --gpus 2` from a plain interpreter: the launcher starts two ranks; on a box without a GPU every rank
    stops at require_gpu() and the launcher hands that failure back (no hang, no silent CPU run)"""
    import subprocess
    import sys
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, IEEE_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode != 0
    lines = r.stdout.decode().strip().splitlines()
    assert len(lines) == 1                         # ONE JSON line: the launcher's error report
    import json
    d = json.loads(lines[0])
    assert d["value"] is None and d["n_gpus"] == 2 and d["rank"] in (0, 1) and d["exit_code"] == r.returncode
    assert "exited with code" in d["error"] and "no CPU fallback" in d["stderr_tail"]
    assert "no CPU fallback" in r.stderr.decode()  # ... and the ranks' stderr still reaches the caller's


def test_bench_self_launch_cannot_hang_silently():
    """ranks that never arrive (the first multi-GPU collective has never met real hardware): after --launch-timeout the
    launcher ends them, prints ONE JSON error line and exits 124"""
    import json
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, IEEE_DIST_BACKEND="gloo", IEEE_BENCH_TEST_HANG_RANK="all")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--launch-timeout", "25"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 124, (r.returncode, r.stderr.decode()[-2000:])
    assert 25 <= time.time() - t0 < 120
    lines = r.stdout.decode().strip().splitlines()
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["value"] is None and d["exit_code"] == 124 and d["error"].startswith("timeout") and d["n_gpus"] == 2


_SLEEPER = """
import os, sys, time
open(os.path.join(%r, "pid%%s" %% os.environ["RANK"]), "w").write(str(os.getpid()))
time.sleep(1e6)
"""


def _pid_alive(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    except PermissionError:
        return True
    try:                                   # a zombie still answers kill(0)
        return open("/proc/%d/stat" % pid).read().split(")")[-1].split()[0] != "Z"
    except OSError:
        return False


def test_launch_timeout_ends_the_ranks_and_returns_124(tmp_path):
    import time
    script = tmp_path / "sleeper.py"
    script.write_text(_SLEEPER % str(tmp_path))
    import sys
    report = {}
    t0 = time.time()
    assert ddp.launch([sys.executable, str(script)], 3, timeout=4.0, grace=2.0, report=report) == 124
    assert time.time() - t0 < 60 and report["timed_out"] and report["rank"] is None and report["code"] == 124
    pids = [int(open(str(tmp_path / ("pid%d" % r))).read()) for r in range(3)]
    assert not any(_pid_alive(p) for p in pids)


def test_a_signal_to_the_launcher_ends_every_rank(tmp_path):
    """SIGTERM to the launching process (a scheduler's kill, `timeout ... python bench.py --gpus 8`): the ranks are terminated
    before the launcher exits with 128 + 15 -- none keeps training or holding a GPU"""
    import signal
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "sleeper.py"
    script.write_text(_SLEEPER % str(tmp_path))
    parent = ("import sys; sys.path.insert(0, %r); from ieee_amd import dist as ddp; "
              "sys.exit(ddp.launch([sys.executable, %r], 3, grace=3.0))" % (root, str(script)))
    p = subprocess.Popen([sys.executable, "-c", parent], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    t_end = time.time() + 120
    while time.time() < t_end and not all((tmp_path / ("pid%d" % r)).exists() and (tmp_path / ("pid%d" % r)).read_text() for r in range(3)):
        time.sleep(0.1)
    pids = [int((tmp_path / ("pid%d" % r)).read_text()) for r in range(3)]
    assert all(_pid_alive(q) for q in pids)
    p.send_signal(signal.SIGTERM)
    assert p.wait(timeout=60) == 128 + signal.SIGTERM
    time.sleep(0.2)
    assert not any(_pid_alive(q) for q in pids)


def test_launch_survives_a_stderr_without_a_descriptor(tmp_path, monkeypatch):
    """argv form with sys.stderr replaced by a capture object (no fileno()): the other ranks' stdout falls back to fd 2"""
    import io
    import sys
    script = tmp_path / "ok.py"
    script.write_text("print('hello')\n")
    monkeypatch.setattr(sys, "stderr", io.StringIO())
    assert ddp.launch([sys.executable, str(script)], 2) == 0


# ---- opt-in bf16 gradient exchange (IEEE_DP_GRAD_DTYPE=bf16): what the halved wire format costs in accuracy ---------------
def _bf16_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank))
    ddp.init_from_env(backend="gloo")
    # every rank's gradient slice: a shared component (the ranks see the same weights) plus a rank-local one, over six
    # decades of magnitude like a real flat gradient (BatchNorm biases ~1e-1 ... layer1 weights ~1e-6)
    g = torch.Generator().manual_seed(100)
    n = 1 << 18
    scale = 10.0 ** (-6.0 * torch.rand(n, generator=g))
    common = torch.randn(n, generator=g)
    local = torch.randn(n, generator=torch.Generator().manual_seed(rank))
    grad = scale * (common + 0.7 * local)
    exact = grad.double()
    dist.all_reduce(exact)                                  # float64 sum of the fp32 slices: the yardstick
    fp32 = grad.clone()
    dist.all_reduce(fp32)
    wire = grad.bfloat16()                                  # ieee_grad_pack_bf16: round to nearest even
    dist.all_reduce(wire)                                   # the collective sums in bf16
    got = wire.float()                                      # ieee_grad_unpack_bf16: exact
    # per element against the sum of magnitudes (a sum that cancels has no relative accuracy in ANY format)
    mags = grad.abs().double()
    dist.all_reduce(mags)
    rel32 = ((fp32.double() - exact).abs() / mags).max().item()
    rel16 = ((got.double() - exact).abs() / mags).max().item()
    cos = torch.nn.functional.cosine_similarity(got.double(), exact, dim=0).item()
    ret[rank] = (rel32, rel16, cos)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_bf16_gradient_allreduce_error_is_bounded(world):
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_bf16_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
        out = dict(ret)
    assert len(out) == world and all(out[r] == out[0] for r in out)          # replicas receive identical sums
    rel32, rel16, cos = out[0]
    assert rel32 < 1e-6
    # one rounding per rank slice (2^-9 each, relative to that slice) + one per partial sum of the reduction
    assert rel16 < 2.0 ** -8 * (1 + world.bit_length()), rel16
    assert cos > 0.99999
