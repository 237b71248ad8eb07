"""GPU: the reference-shaped engine surface.  Fused train step (native forward, 18-head CE, 3M,
backward, fused SGD) against the oracle's CPU restatement of the reference step; the autograd path with
torch.optim.SGD against the fused path; and the device-resident evaluation pipeline."""
import numpy as np
import pytest
import torch

from oracle import model as om
from tests.util_model import C, generated_state, images

pytestmark = pytest.mark.gpu


class FakeDM(object):
    num_train_pids = C
    sources = ["synthetic"]

    def __init__(self, test_loader=None):
        self.train_loader = []
        self.test_loader = test_loader or {}


def make(seed, fused=True, dtype=torch.float32):
    from ieee_amd.engine import Image3MEngine
    from ieee_amd.models import build_model
    from ieee_amd.optim import build_optimizer
    m = build_model("ieee3modalPart", num_classes=C, loss="margin", pretrained=False, compute_dtype=dtype)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = generated_state(shapes, seed)
    m.load_state_dict(sd)
    opt = build_optimizer(m, optim="sgd", lr=1e-3, weight_decay=5e-4, momentum=0.9, fused=fused)
    eng = Image3MEngine(FakeDM(), m, opt, margin=1, weight_m=1, weight_x=1, use_gpu=True, label_smooth=True)
    m.train()
    return eng, m, sd


def batch(B, seed):
    xs = images(B, seed)
    pids = torch.arange(B) // 4
    return {"img": xs, "pid": pids, "camid": pids * 0, "impath": "", "timeid": pids * 0}


def test_fused_step_matches_oracle_step():
    B, seed = 8, 3
    eng, m, sd = make(seed)
    s = eng.forward_backward(batch(B, seed))
    assert set(s) == {"loss", "LossX", "LossM", "accR", "lossR", "accN", "lossN", "accT", "lossT"}
    assert torch.is_tensor(s["LossM"]) and s["LossM"].dim() == 0          # the reference returns a 0-d tensor here
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    ref, grads, new_sd, _ = om.train_step(sd, images(B, seed), torch.arange(B) // 4, C)
    for k in ("loss", "LossX", "lossR", "lossN", "lossT", "accR", "accN", "accT"):
        assert abs(float(s[k]) - ref[k]) <= 1e-4 * max(1.0, abs(ref[k])), k
    assert abs(float(s["LossM"]) - ref["LossM"]) < 1e-4
    # parameters after one SGD-nesterov step (lr 1e-3): update = lr*(g + wd*p)*(1+momentum)
    mine = m.state_dict()
    worst = 0.0
    for k, v in new_sd.items():
        if k.endswith("num_batches_tracked") or "running_" in k:
            continue
        upd_ref = (new_sd[k] - sd[k])
        upd_my = (mine[k].cpu() - sd[k])
        denom = upd_ref.abs().max().item()
        if denom < 1e-12:
            assert upd_my.abs().max().item() < 1e-9, k            # conv_value: untouched
            continue
        worst = max(worst, (upd_my - upd_ref).abs().max().item() / denom)
    assert worst < 0.25      # per-tensor max error relative to the largest update; reference noise floor ~0.14
    # conv_query receives zero gradients but is still decayed; conv_value is untouched
    q, v = "REM.0.conv_query.weight", "REM.0.conv_value.weight"
    torch.testing.assert_close(mine[q].cpu(), new_sd[q], rtol=1e-6, atol=1e-9)
    assert torch.equal(mine[v].cpu(), sd[v])


def test_autograd_path_equals_fused_path():
    B, seed = 8, 4
    eng_f, m_f, _ = make(seed, fused=True)
    eng_a, m_a, _ = make(seed, fused=False)
    assert type(eng_a.optimizer).__name__ == "SGD"
    b = batch(B, seed)
    s_f = eng_f.forward_backward({k: ([x.clone() for x in v] if k == "img" else v) for k, v in b.items()})
    s_a = eng_a.forward_backward(b)
    for k in ("loss", "LossX", "lossR", "accT"):
        assert abs(float(s_f[k]) - float(s_a[k])) < 1e-4 * max(1, abs(float(s_a[k])))
    sd_f, sd_a = m_f.state_dict(), m_a.state_dict()
    for k in sd_f:
        if sd_f[k].is_floating_point():
            assert (sd_f[k] - sd_a[k]).abs().max().item() <= 1e-7 + 1e-5 * sd_a[k].abs().max().item(), k
    # two more steps exercise the momentum buffers on both sides
    for _ in range(2):
        eng_f.forward_backward(batch(B, seed))
        eng_a.forward_backward(batch(B, seed))
    for k in ("backbone.0.conv1.weight", "classifier_T.5.weight", "REM.1.conv_query.weight"):
        a, f = m_a.state_dict()[k], m_f.state_dict()[k]
        assert (a - f).abs().max().item() <= 1e-2 * (a - _orig(k, seed)).abs().max().item() + 1e-8, k


def _orig(k, seed):
    from ieee_amd._spec import state_spec
    shapes = {n: s for n, s, _ in state_spec(C)}
    return generated_state({k: shapes[k]}, seed)[k].cuda()


def test_softmax_engine_and_eval_pipeline(capsys):
    from ieee_amd.engine import MultiModalImageSoftmaxEngine
    from ieee_amd.models import build_model
    from ieee_amd.optim import build_optimizer
    m = build_model("ieee3modalPart", num_classes=C, loss="softmax", pretrained=False, compute_dtype=torch.bfloat16)
    opt = build_optimizer(m, optim="sgd", lr=1e-3)
    # tiny synthetic query/gallery loaders in the reference's batch-dict format
    def loader(n, seed, cam):
        xs = images(n, seed)
        return [{"img": [x[i:i + 4] for x in xs], "pid": torch.arange(i, min(i + 4, n)) % 5,
                 "camid": torch.full((min(4, n - i),), cam), "impath": "", "timeid": torch.zeros(min(4, n - i))}
                for i in range(0, n, 4)]
    dm = FakeDM({"synthetic": {"query": loader(8, 1, 0), "gallery": loader(12, 2, 1)}})
    eng = MultiModalImageSoftmaxEngine(dm, m, opt, use_gpu=True)
    m.train()
    s = eng.forward_backward(batch(8, 1))
    assert set(s) == {"loss_all", "loss_R", "acc_R", "loss_N", "acc_N", "loss_T", "acc_T"}
    assert np.isfinite(s["loss_all"])
    rank1_map = eng.test(ranks=[1, 5, 10])      # 12 gallery rows: Rank-20 does not exist (the reference raises there too)
    out = capsys.readouterr().out
    assert "mAP:" in out and "Rank-1" in out and 0.0 <= rank1_map <= 1.0


def test_fused_adam_engine_step_matches_torch_adam_on_the_same_gradients():
    """build_optimizer(optim='amsgrad') returns FusedAdam for the native model; one engine step must move the
    parameters exactly like torch.optim.Adam(amsgrad=True) fed with the gradients that step left in the flat buffer"""
    from ieee_amd.engine import Image3MEngine
    from ieee_amd.models import build_model
    from ieee_amd.optim import FusedAdam, build_optimizer
    from tests.util_model import generated_state, images
    m = build_model("ieee3modalPart", num_classes=C, loss="margin", pretrained=False, compute_dtype=torch.float32)
    m.load_state_dict(generated_state({k: tuple(v.shape) for k, v in m.state_dict().items()}, 5))
    opt = build_optimizer(m, optim="amsgrad", lr=3e-4, weight_decay=5e-4)
    assert isinstance(opt, FusedAdam)
    eng = Image3MEngine(FakeDM(), m, opt, margin=1, use_gpu=True)
    m.train()
    before = m._flat_params.clone()
    B = 8
    pids = torch.arange(B) // 4
    eng.forward_backward({"img": images(B, 5), "pid": pids, "camid": pids * 0, "impath": "", "timeid": pids * 0})
    grads = m._flat_grads.clone()
    ref = torch.nn.Parameter(before.clone())
    ref.grad = grads.clone()
    torch.optim.Adam([ref], lr=3e-4, betas=(0.9, 0.99), weight_decay=5e-4, amsgrad=True).step()
    # parameters that receive no gradient (REM.conv_value) are not part of any trainable run and must not move
    moved = torch.zeros_like(before, dtype=torch.bool)
    for a, b in m.trainable_runs():
        moved[a:b] = True
    torch.testing.assert_close(m._flat_params[moved], ref.detach()[moved], rtol=1e-5, atol=1e-7)
    assert torch.equal(m._flat_params[~moved], before[~moved])


def test_multi_stream_step_is_bitwise_reproducible():
    """the executor overlaps weight gradients, weight packing, the downsample branches and most of the optimizer update
    on side streams; every kernel is deterministic, so two runs of the same 4 steps from the same state must end in
    bit-identical parameters, momentum buffers and running statistics -- a missing stream dependency would show up
    here (and only sporadically anywhere else)"""
    from ieee_amd.engine import Image3MEngine
    from ieee_amd.models import build_model
    from ieee_amd.optim import build_optimizer
    from tests.util_model import generated_state, images
    B = 16
    pids = torch.arange(B) // 4
    batches = [{"img": images(B, 30 + i), "pid": pids, "camid": pids * 0, "impath": "", "timeid": pids * 0} for i in range(2)]
    finals = []
    for run in range(2):
        m = build_model("ieee3modalPart", num_classes=C, loss="margin", pretrained=False, compute_dtype=torch.bfloat16)
        m.load_state_dict(generated_state({k: tuple(v.shape) for k, v in m.state_dict().items()}, 9))
        opt = build_optimizer(m, optim="sgd", lr=1e-2, weight_decay=5e-4, momentum=0.9)
        eng = Image3MEngine(FakeDM(), m, opt, margin=1, use_gpu=True)
        m.train()
        losses = [eng.forward_backward(batches[i % 2])["loss"] for i in range(4)]
        torch.cuda.synchronize()
        finals.append((m._flat_params.clone(), m._flat_buffers.clone(), opt.momentum_buffer().clone(), losses))
        del eng, opt, m
    assert finals[0][3] == finals[1][3]
    for a, b in zip(finals[0][:3], finals[1][:3]):
        assert torch.equal(a, b)
    assert all(l == l for l in finals[0][3])          # no NaN


def test_optimizer_shadow_feeds_the_forward_the_bits_of_the_packed_operands():
    """FusedSGD writes the bf16 image of the updated parameters beside them (ieee_sgd_nesterov_step_shadow) and the bf16
    training forward reads the 1x1 convolutions' GEMM operands from that shadow instead of packing them: the conversion is
    the packing's, so 6 steps end in BIT-IDENTICAL parameters / momentum / running statistics / losses with the shadow on
    and off -- including a step after somebody else wrote parameters (an in-place edit, a load_state_dict: the shadow is
    refreshed from torch's version counter) and a second forward without an optimizer step in between."""
    from ieee_amd.engine import Image3MEngine
    from ieee_amd.models import build_model
    from ieee_amd.optim import build_optimizer
    from tests.util_model import generated_state, images
    B = 8
    pids = torch.arange(B) // 4
    batches = [{"img": images(B, 40 + i), "pid": pids, "camid": pids * 0, "impath": "", "timeid": pids * 0} for i in range(2)]
    finals = []
    for shadow in (False, True):
        m = build_model("ieee3modalPart", num_classes=C, loss="margin", pretrained=False, compute_dtype=torch.bfloat16)
        state = generated_state({k: tuple(v.shape) for k, v in m.state_dict().items()}, 9)
        m.load_state_dict(state)
        opt = build_optimizer(m, optim="sgd", lr=1e-2, weight_decay=5e-4, momentum=0.9)
        assert m._shadow_enabled is True            # FusedSGD + bf16 model: on by default
        m._shadow_enabled = shadow
        eng = Image3MEngine(FakeDM(), m, opt, margin=1, use_gpu=True)
        m.train()
        losses = [float(eng.forward_backward(batches[i % 2])["loss"]) for i in range(2)]
        if shadow:
            assert m._shadow_key == (m._flat_params._version, m._native_epoch)          # kept current by the optimizer
            torch.cuda.synchronize()
            assert torch.equal(m._flat_shadow, m._flat_params.to(torch.bfloat16))
        with torch.no_grad():
            dict(m._param_items)["backbone.0.layer3.1.conv1.weight"].mul_(1.5)          # a 1x1 conv the forward reads from the shadow
        losses += [float(eng.forward_backward(batches[i % 2])["loss"]) for i in range(2)]
        m.load_state_dict({k: v * 0.5 if k.endswith("layer4.0.conv3.weight") else v for k, v in m.state_dict().items()})
        out = m([x.cuda() for x in batches[0]["img"]])                                   # a training forward without a step behind it
        losses.append(float(sum(o.float().abs().sum() for o in out[3:])))
        losses += [float(eng.forward_backward(batches[i % 2])["loss"]) for i in range(2)]
        torch.cuda.synchronize()
        finals.append((m._flat_params.clone(), m._flat_buffers.clone(), opt.momentum_buffer().clone(), losses))
        del eng, opt, m
    assert finals[0][3] == finals[1][3], (finals[0][3], finals[1][3])
    for a, b in zip(finals[0][:3], finals[1][:3]):
        assert torch.equal(a, b)
    assert all(l == l for l in finals[0][3])


def test_eval_cache_follows_parameter_changes():
    """consecutive eval forwards reuse the packed weights / BN scale-shift; any change of parameters or running
    statistics (torch in-place ops, load_state_dict, a train step, invalidate_eval_cache after a .data write) must
    be picked up by the next eval forward"""
    from ieee_amd.models import build_model
    from ieee_amd.optim import build_optimizer
    from ieee_amd.engine import Image3MEngine
    from tests.util_model import generated_state, images
    B = 8
    m = build_model("ieee3modalPart", num_classes=C, loss="margin", pretrained=False, compute_dtype=torch.float32)
    state = generated_state({k: tuple(v.shape) for k, v in m.state_dict().items()}, 3)
    m.load_state_dict(state)
    m.eval()
    x = [t.cuda() for t in images(B, 3)]

    def fresh_reference():
        r = build_model("ieee3modalPart", num_classes=C, loss="margin", pretrained=False, compute_dtype=torch.float32)
        r.load_state_dict(m.state_dict())
        r.eval()
        with torch.no_grad():
            return r(x).clone()

    with torch.no_grad():
        f0 = m(x).clone()
        f1 = m(x).clone()                                   # cached path
    assert torch.equal(f0, f1) and torch.equal(f0, fresh_reference())
    with torch.no_grad():
        m.backbone[0].layer1[0].conv1.weight.mul_(1.5)      # in-place op on a view: version counter
        f2 = m(x).clone()
    assert not torch.equal(f2, f0) and torch.equal(f2, fresh_reference())
    with torch.no_grad():
        m.backbone[1].bn1.running_var.add_(0.25)            # a buffer
        f3 = m(x).clone()
    assert not torch.equal(f3, f2) and torch.equal(f3, fresh_reference())
    m.backbone[2].layer4[2].conv3.weight.data.mul_(0.5)     # .data write: invisible to the version counters
    m.invalidate_eval_cache()
    with torch.no_grad():
        f4 = m(x).clone()
    assert not torch.equal(f4, f3) and torch.equal(f4, fresh_reference())
    # a train step in between (native writers) and back to eval
    opt = build_optimizer(m, optim="sgd", lr=1e-2, weight_decay=0.0, momentum=0.0)
    eng = Image3MEngine(FakeDM(), m, opt, margin=1, use_gpu=True)
    m.train()
    pids = torch.arange(B) // 4
    eng.forward_backward({"img": images(B, 4), "pid": pids, "camid": pids * 0, "impath": "", "timeid": pids * 0})
    m.eval()
    with torch.no_grad():
        f5 = m(x).clone()
    assert not torch.equal(f5, f4) and torch.equal(f5, fresh_reference())


def test_profile_passes_time_the_launches_without_changing_the_step():
    """ieee_net_profile: mode 2 (every forward / dgrad launch carries an event pair as its own start / stop signals, the
    executor's streams stay on) and mode 1 (event records around every launch on ONE ordered stream) both report the 109
    forward + dgrad and the 55 weight-gradient launches of a step with positive durations and the algorithmic FLOPs of the
    net, and a profiled step leaves bit-identical parameters (bench.py's roofline passes rest on both)"""
    import ctypes
    from ieee_amd import _lib
    B = 8
    data = batch(B, 5)
    finals, reports = [], {}
    for mode in (0, 2, 1):
        eng, m, _ = make(5, dtype=torch.bfloat16)
        eng.forward_backward(data)                      # (first step: stream / workspace set-up, self-check)
        net = m.native_net(B, 256, 128)
        lib = _lib.load()
        if mode:
            _lib.check(lib.ieee_net_profile(net.handle, mode, None))
        s = eng.forward_backward(data)
        if mode:
            out = (ctypes.c_double * 6)()
            _lib.check(lib.ieee_net_profile(net.handle, 0, out))
            reports[mode] = list(out)
        torch.cuda.synchronize()
        finals.append((float(s["loss"]), m._flat_params.clone()))
        del eng, m, net
        torch.cuda.empty_cache()
    for mode, (g_ms, g_fl, g_n, w_ms, w_fl, w_n) in reports.items():
        assert (g_n, w_n) == (109, 55), (mode, g_n, w_n)
        assert 0.01 < g_ms < 50 and 0.01 < w_ms < 50, (mode, g_ms, w_ms)
        # 61.06 / 30.76 GFLOP per triple (forward + dgrad without the stems' dgrad; weight gradients)
        assert abs(g_fl / B / 61.0617e9 - 1) < 1e-3 and abs(w_fl / B / 30.762e9 - 1) < 1e-3, (mode, g_fl, w_fl)
    assert finals[0][0] == finals[1][0] == finals[2][0]
    assert torch.equal(finals[0][1], finals[1][1]) and torch.equal(finals[0][1], finals[2][1])
