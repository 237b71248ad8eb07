"""CPU: host logic of the engine that needs no device -- the guard for reference-side loaders that fork() the training
process (torchreid/data/datamanager.py:214-229 builds DataLoader(num_workers=workers) with the default start method;
scripts/default_config.py:20: workers = 1)."""
import warnings

import pytest
import torch

from ieee_amd.engine import Engine


class _DS(torch.utils.data.Dataset):
    def __len__(self):
        return 4

    def __getitem__(self, i):
        return i


class _DM(object):
    num_train_pids = 3
    sources = ["synthetic"]
    test_loader = {}

    def __init__(self, loader):
        self.train_loader = loader


def _engine(loader, monkeypatch, live=True):
    monkeypatch.setattr(torch.cuda, "is_available", lambda: live)
    monkeypatch.setattr(torch.cuda, "is_initialized", lambda: live)
    return Engine(_DM(loader), use_gpu=False)


def test_forking_dataloader_beside_a_live_hip_context_warns_once(monkeypatch):
    eng = _engine(torch.utils.data.DataLoader(_DS(), batch_size=2, num_workers=1, multiprocessing_context="fork"), monkeypatch)
    with pytest.warns(RuntimeWarning, match="forkserver") as rec:
        eng._warn_forking_loader()
    assert "364 ms" in str(rec[0].message) and "1 worker" in str(rec[0].message)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        eng._warn_forking_loader()                      # once per engine


def test_default_start_method_on_linux_counts_as_fork(monkeypatch):
    import multiprocessing
    if multiprocessing.get_start_method(allow_none=True) not in (None, "fork"):
        pytest.skip("this interpreter's default start method is not fork")
    eng = _engine(torch.utils.data.DataLoader(_DS(), batch_size=2, num_workers=2), monkeypatch)
    with pytest.warns(RuntimeWarning, match="2 worker"):
        eng._warn_forking_loader()


@pytest.mark.parametrize("kw,live", [(dict(num_workers=0), True), (dict(num_workers=1, multiprocessing_context="forkserver"), True),
                                     (dict(num_workers=1, multiprocessing_context="spawn"), True),
                                     (dict(num_workers=1, multiprocessing_context="fork"), False)])
def test_no_warning_without_the_hazard(monkeypatch, kw, live):
    eng = _engine(torch.utils.data.DataLoader(_DS(), batch_size=2, **kw), monkeypatch, live=live)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        eng._warn_forking_loader()
    # any other iterable (a list of batches, ieee_amd.data's DeviceLoader) is not a forking loader
    eng = _engine([{"img": None}], monkeypatch)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        eng._warn_forking_loader()


# ---- range guard of the fixed-point BatchNorm totals: what the engine does with the step's report words ---------------------
class _FakeNet(object):
    def __init__(self):
        self.totals_on = True

    def set_bn_totals(self, on):
        self.totals_on = bool(on)


class _FakeModel(object):
    _bn_totals_off = False


def _fused_engine():
    from ieee_amd.engine import _FusedStepMixin
    eng = _FusedStepMixin()
    eng.model = _FakeModel()
    return eng


def test_clamped_batchnorm_tile_degrades_when_the_step_was_skipped_and_raises_otherwise(monkeypatch):
    from ieee_amd._lib import IeeeAmdError
    monkeypatch.delenv("IEEE_BN_STRICT", raising=False)
    # not guarded (data parallel / another optimizer: the update has been applied): raise, and name the switch
    eng, net = _fused_engine(), _FakeNet()
    with pytest.raises(IeeeAmdError, match="IEEE_BN_TOTALS_TILES=0") as e:
        eng._check_bn_range(net, 7, (1, 0, 0, 0), False)
    assert "forward" in str(e.value) and "backward (" not in str(e.value) and "step 7" in str(e.value) and "HAS BEEN APPLIED" in str(e.value)
    with pytest.raises(IeeeAmdError, match="backward"):
        eng._check_bn_range(net, 8, (0, 1, 1, 1), False)
    assert net.totals_on and not eng.model._bn_totals_off
    # guarded (the device skipped the step): switch to the partial-sum path, warn ONCE with the step, go on
    with pytest.warns(RuntimeWarning, match=r"of step 9.*was SKIPPED.*partial-sum path"):
        eng._check_bn_range(net, 9, (1, 0, 0, 0), True)
    assert not net.totals_on and eng.model._bn_totals_off
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        eng._check_bn_range(net, 10, (0, 1, 0, 0), True)         # a step queued behind it that clamped too: skipped as well, no second warning
    # IEEE_BN_STRICT=1: raise even when the step was skipped
    monkeypatch.setenv("IEEE_BN_STRICT", "1")
    with pytest.raises(IeeeAmdError, match="IEEE_BN_STRICT=1"):
        _fused_engine()._check_bn_range(_FakeNet(), 3, (1, 0, 0, 0), True)


def test_half_range_total_warns_once_and_healthy_steps_are_silent():
    eng = _fused_engine()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        eng._check_bn_range(_FakeNet(), 1, (0, 0, 0, 0), True)
        eng._check_bn_range(None, 1, (0, 0, 0, 0), False)          # the generic (autograd) step has no executor
    with pytest.warns(UserWarning, match="beyond half the range"):
        eng._check_bn_range(_FakeNet(), 2, (0, 0, 1, 0), True)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        eng._check_bn_range(_FakeNet(), 3, (0, 0, 0, 1), True)     # warned once per engine

