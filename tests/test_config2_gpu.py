"""GPU: the TIMED configuration under the checker (BASELINE config 2: B = 64 triples, 256x128, bf16).  The tile choice
(128x128 vs 128x64 at <= 512 workgroups), the weight-gradient split-K and the XCD tile grouping all depend on the batch
size, so the kernel instantiations bench.py times are exercised here at exactly its shapes: every distinct conv layer
shape (forward, dgrad, wgrad, fused BatchNorm sums), the per-stage drift of the whole bf16 forward against the fp32
oracle, the fp32 parity mode against the oracle's CPU forward at 1e-3, and the bf16 train step's loss."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from ieee_amd._spec import state_spec
from oracle import model as om
from tests.util_model import C, generated_state, images

pytestmark = pytest.mark.gpu
B = 64

# distinct conv shapes of one ResNet-50 stream + the CIM at 256x128: Ci, Co, R, stride, Hin, Win
LAYERS = [(64, 64, 1, 1, 64, 32), (64, 64, 3, 1, 64, 32), (64, 256, 1, 1, 64, 32), (256, 64, 1, 1, 64, 32),
          (256, 128, 1, 1, 64, 32), (128, 128, 3, 2, 64, 32), (128, 512, 1, 1, 32, 16), (256, 512, 1, 2, 64, 32),
          (512, 128, 1, 1, 32, 16), (128, 128, 3, 1, 32, 16), (512, 256, 1, 1, 32, 16), (256, 256, 3, 2, 32, 16),
          (256, 1024, 1, 1, 16, 8), (512, 1024, 1, 2, 32, 16), (1024, 256, 1, 1, 16, 8), (256, 256, 3, 1, 16, 8),
          (1024, 512, 1, 1, 16, 8), (512, 512, 3, 1, 16, 8), (512, 2048, 1, 1, 16, 8), (1024, 2048, 1, 1, 16, 8),
          (2048, 512, 1, 1, 16, 8), (2048, 2048, 1, 1, 16, 8)]


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


@pytest.mark.parametrize("layer", LAYERS, ids=lambda l: "%d-%d_k%d_s%d_%dx%d" % l)
def test_conv_kernels_at_config2_shapes(layer):
    """bf16 forward (+ fused BN sums) / dgrad / wgrad of one layer shape at G = 3 modalities x N = 64, against torch's
    fp32 convolution on the SAME bf16-rounded operands (exact products, fp32 accumulation on both sides: the remaining
    difference is the summation order and the one bf16 rounding of each stored output)"""
    from ieee_amd import _lib as L, _ops
    lib = L.require_gpu()
    Ci, Co, R, stride, H, W = layer
    pad = R // 2
    G = 3
    g = torch.Generator(device="cuda").manual_seed(Ci * 7 + Co + R)
    dt = torch.bfloat16
    x = torch.randn(G, B, H, W, Ci, generator=g, device="cuda").to(dt)
    w = (torch.randn(G, Co, Ci, R, R, generator=g, device="cuda") * (2.0 / (Ci * R * R)) ** 0.5).to(dt).float()
    Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    dy = torch.randn(G, B, Ho, Wo, Co, generator=g, device="cuda").to(dt)
    wp, wpd = _ops.pack_conv_weight(w, dt, 0), _ops.pack_conv_weight(w, dt, 1)
    rb = lib.ieee_conv2d_fwd_stats_rblocks(B, Ho, Wo)
    part = torch.zeros(G, 2, Co, rb, device="cuda")
    y = torch.empty(G, B, Ho, Wo, Co, device="cuda", dtype=dt)
    L.check(lib.ieee_conv2d_fwd(L.ptr(x), L.ptr(wp), L.ptr(y), L.IEEE_BF16, G, B, H, W, Ci, Co, R, R, stride, pad,
                                x[0].numel(), wp.stride(0), y[0].numel(), L.ptr(part), L.stream()))
    dx = _ops.conv2d_dgrad(dy, wpd, (H, W), Ci, R, R, stride, pad)
    dw = _ops.conv2d_wgrad(dy, x, R, R, stride, pad)
    # the same gradient with the split-K fold inside the launch (ieee_conv2d_wgrad_fold: what the executor runs): same
    # tolerance against autograd below, bit-identical from run to run, `accumulate` adds to what is there
    dwf = _ops.conv2d_wgrad(dy, x, R, R, stride, pad, fold=True)
    for _ in range(3):
        assert torch.equal(_ops.conv2d_wgrad(dy, x, R, R, stride, pad, fold=True), dwf), "fold: not reproducible"
    dwf2 = _ops.conv2d_wgrad(dy, x, R, R, stride, pad, out=dwf.clone(), accumulate=True, fold=True)
    assert torch.equal(dwf2, dwf + dwf), "fold: accumulate"
    assert int(_ops._fold_tickets(dy.device).abs().sum()) == 0, "fold: a launch left its tickets behind"
    for i in range(G):
        xi = x[i].float().permute(0, 3, 1, 2).requires_grad_(True)
        wi = w[i].clone().requires_grad_(True)
        yr = F.conv2d(xi, wi, None, stride, pad)
        yr.backward(dy[i].float().permute(0, 3, 1, 2))
        yr = yr.detach().permute(0, 2, 3, 1)
        assert _rel(y[i].float(), yr) < 3e-3, "forward"                 # rms of one bf16 rounding = 2^-9 / sqrt(3) = 1.1e-3
        assert float((y[i].float() - yr).abs().max()) <= 2 ** -7 * float(yr.abs().max()) + 1e-3
        assert _rel(dx[i].float(), xi.grad.permute(0, 2, 3, 1)) < 3e-3, "dgrad"
        assert _rel(dw[i], wi.grad) < 2e-4, "wgrad"                     # fp32 output: summation order only
        assert _rel(dwf[i], wi.grad) < 2e-4, "wgrad (in-launch fold)"
        yf = y[i].float().view(-1, Co)
        torch.testing.assert_close(part[i, 0].sum(-1), yf.sum(0), rtol=2e-4, atol=0.5)
        torch.testing.assert_close(part[i, 1].sum(-1), (yf * yf).sum(0), rtol=2e-4, atol=0.5)


def _model(dtype, seed):
    from ieee_amd.models import build_model
    m = build_model("ieee3modalPart", num_classes=C, loss="margin", pretrained=False, use_gpu=True, compute_dtype=dtype)
    m.load_state_dict(generated_state({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed))
    return m.train()


def test_b64_bf16_stage_drift_not_worse_than_stock_bf16():
    """the B = 64 twin of tests/test_model_gpu.py::test_bf16_mode_drift_not_worse_than_stock_bf16: at the stem and at the
    end of every stage the native bf16 forward must be no further from the fp32 oracle (stock torch fp32 ops, here on the
    device) than stock torch bf16 autocast is, x 1.25"""
    seed = 2
    m = _model(torch.bfloat16, seed)
    xs = [x.cuda() for x in images(B, seed)]
    out = m(xs)
    net = list(m._nets.values())[0]
    sd = {k: v.cuda() for k, v in generated_state({k: s for k, s, _ in state_spec(C)}, seed).items()}
    t32, t16 = {}, {}
    with torch.no_grad():
        o32 = om.forward({k: v.clone() for k, v in sd.items()}, xs, True, taps=t32)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            o16 = om.forward({k: v.clone() for k, v in sd.items()}, xs, True, taps=t16)
    rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())
    # (the training forward never writes the stem's full-resolution activation: its max-pooled form is what exists)
    stages = [("stem", "pool")] + [("layer%d" % l, "backbone.{m}.layer%d.%d.conv3.a" % (l, last))
                                  for l, last in ((1, 2), (2, 3), (3, 5), (4, 2))]
    pooled = lambda t, tap: torch.nn.functional.max_pool2d(t.float(), 3, 2, 1).to(t.dtype) if tap == "stem" else t
    for tap, name in stages:
        ref = torch.stack([pooled(t32["backbone.%d.%s" % (i, tap)], tap) for i in range(3)]).permute(0, 1, 3, 4, 2)
        stock = torch.stack([pooled(t16["backbone.%d.%s" % (i, tap)], tap) for i in range(3)]).permute(0, 1, 3, 4, 2)
        mine = net.tensor(name).view(ref.shape)
        e_mine, e_stock = rel(mine, ref), rel(stock, ref)
        print("%-8s native bf16 %.3e   stock torch bf16 %.3e" % (tap, e_mine, e_stock))
        assert e_mine <= 1.25 * e_stock + 1e-3, tap
    f_mine, f_stock, f_ref = torch.stack(list(out[3:])), torch.stack(list(o16[3:])), torch.stack(list(o32[3:]))
    assert rel(f_mine, f_ref) <= 1.25 * rel(f_stock, f_ref) + 1e-3
    pids = (torch.arange(B) // 4).cuda()
    loss, _ = om.losses(out, pids, C)
    loss32, _ = om.losses(o32, pids, C)
    assert abs(float(loss) - float(loss32)) / float(loss32) < 0.02
    loss.backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)


def test_b64_fp32_parity_mode_matches_oracle_cpu_forward():
    """fp32 parity mode at B = 64: logits and normalised features within 1e-3 of the oracle's CPU forward
    (north_star's contract), losses within 1e-4"""
    seed = 12
    m = _model(torch.float32, seed)
    xs = images(B, seed)
    out = m([x.cuda() for x in xs])
    sd = generated_state({k: s for k, s, _ in state_spec(C)}, seed)
    torch.set_num_threads(max(1, min(32, torch.get_num_threads())))
    with torch.no_grad():
        ref = om.forward(sd, xs, True)
    for mine, theirs in zip(out[:3], ref[:3]):
        for a, b in zip(mine, theirs):
            assert float((a.detach().cpu() - b).abs().max()) < 1e-3
    for a, b in zip(out[3:], ref[3:]):
        assert float((a.detach().cpu() - b).abs().max()) < 1e-3
    pids = torch.arange(B) // 4
    _, s_ref = om.losses(ref, pids, C)
    _, s_my = om.losses([[t.detach().cpu() for t in o] for o in out[:3]] + [t.detach().cpu() for t in out[3:]], pids, C)
    for k in ("loss", "LossX", "LossM", "lossR", "lossN", "lossT"):
        assert abs(float(s_my[k]) - float(s_ref[k])) <= 1e-4 * max(1.0, abs(float(s_ref[k]))), k


def test_b64_bf16_engine_step_is_the_bench_step():
    """exactly what bench.py times -- Image3MEngine.forward_backward, fused path, B = 64, bf16, randn images -- gives a
    finite loss within 2 % of the fp32 parity mode on the same batch, finite parameters and the same loss when repeated
    from the same state (the multi-stream step is deterministic)"""
    from ieee_amd.engine import Image3MEngine
    from ieee_amd.optim import build_optimizer

    class DM(object):
        num_train_pids = C
        train_loader, test_loader, sources = [], {}, ["synthetic"]
    g = torch.Generator().manual_seed(0)
    imgs = [torch.randn(B, 3, 256, 128, generator=g) for _ in range(3)]
    pids = torch.arange(B) // 4
    losses = {}
    for tag, dtype in (("bf16_a", torch.bfloat16), ("bf16_b", torch.bfloat16), ("fp32", torch.float32)):
        m = _model(dtype, 0)
        eng = Image3MEngine(DM(), m, build_optimizer(m, optim="sgd", lr=1e-3, weight_decay=5e-4, momentum=0.9), margin=1,
                            use_gpu=True)
        s = eng.forward_backward({"img": imgs, "pid": pids, "camid": pids * 0, "impath": "", "timeid": pids * 0})
        torch.cuda.synchronize()
        assert torch.isfinite(m._flat_params).all() and torch.isfinite(m._flat_grads).all()
        losses[tag] = float(s["loss"])
        del eng, m
    assert losses["bf16_a"] == losses["bf16_b"]
    assert abs(losses["bf16_a"] - losses["fp32"]) / losses["fp32"] < 0.02, losses


def test_wgrad_fold_hand_off_under_load_and_with_warm_caches():
    """The in-launch split-K fold hands fp32 tiles from workgroup to workgroup (write-through stores, ticket, acquire): a stale
    line read by a reducer would show as a wrong gradient only sometimes.  So: alternate two layer shapes that share ONE
    workspace and ONE ticket block (what the executor does), keep a bandwidth-heavy kernel running on a second stream, and
    require every repetition to return the bits of the first."""
    from ieee_amd import _ops
    dt = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(5)
    shapes = [(256, 1024, 1, 16, 8), (64, 256, 1, 64, 32), (256, 256, 3, 16, 8), (128, 128, 3, 32, 16), (512, 2048, 1, 16, 8)]
    ops = []
    for Ci, Co, R, H, W in shapes:
        x = torch.randn(3, B, H, W, Ci, generator=g, device="cuda").to(dt)
        dy = torch.randn(3, B, H, W, Co, generator=g, device="cuda").to(dt)
        ops.append((x, dy, R))
    first = [_ops.conv2d_wgrad(dy, x, R, R, 1, R // 2, fold=True).clone() for x, dy, R in ops]
    ref = [_ops.conv2d_wgrad(dy, x, R, R, 1, R // 2) for x, dy, R in ops]
    for a, b in zip(first, ref):
        assert _rel(a, b) < 1e-5        # same partial tiles, another association of the fp32 adds
    big = torch.empty(256 << 20, device="cuda")
    side = torch.cuda.Stream()
    for rep in range(6):
        with torch.cuda.stream(side):
            for _ in range(8):
                big.add_(1.0)          # 2 GB of traffic per pass beside the folds
        for (x, dy, R), want in zip(ops, first):
            got = _ops.conv2d_wgrad(dy, x, R, R, 1, R // 2, fold=True)
            assert torch.equal(got, want), "fold: hand-off returned different bits (rep %d)" % rep
    torch.cuda.synchronize()
    assert int(_ops._fold_tickets(big.device).abs().sum()) == 0


@pytest.mark.parametrize("env", ["IEEE_GATHER_DUAL=4 IEEE_GATHER_STAGGER=3 IEEE_STEM_WALK=4 IEEE_WGRAD_PIPE=6",
                                 "IEEE_WGRAD_CHAIN=1 IEEE_HEAD_PAIRS=0 IEEE_SGD_SHADOW=0", "IEEE_WGRAD_FOLD=1"])
def test_optional_kernel_forms_stay_correct(env):
    """the kernel forms round 6 built, measured and left OFF (dual-issue k-loops of both GEMM cores, phase stagger, persistent stem,
    chained / folded weight-gradient reductions) and the two it switched ON (head pairs, parameter shadow; here switched off) are options of the
    product: every layer shape of the timed configuration and one engine step must still pass under each.  The switches are read
    once per process, hence a child interpreter."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = dict(os.environ)
    for kv in env.split():
        k, v = kv.split("=")
        child[k] = v
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        os.path.join(root, "tests", "test_config2_gpu.py"), "-k", "conv_kernels_at_config2_shapes or b64_bf16_engine_step or direct_stem",
                        os.path.join(root, "tests", "test_conv_gpu.py") + "::test_direct_stem_persistent_over_four_tiles"],
                       cwd=root, env=child, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=570)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
