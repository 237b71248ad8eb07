"""MetricMeter / AverageMeter (reference utils/avgmeter.py:8-73) and the deferred loss summary of the fused step."""
import json

import pytest
import torch

from ieee_amd.meters import AverageMeter, DeferredSummary, MetricMeter


def test_average_meter_matches_the_reference_arithmetic():
    m = AverageMeter()
    assert m.avg == 0
    for v, n in ((2.0, 1), (4.0, 3)):
        m.update(v, n)
    assert (m.val, m.sum, m.count) == (4.0, 14.0, 4) and m.avg == 3.5


def test_metric_meter_reads_tensors_and_rejects_non_dicts():
    mm = MetricMeter(delimiter=" | ")
    mm.update({"loss": 2.0, "LossM": torch.tensor(4.0)})
    mm.update({"loss": 4.0, "LossM": torch.tensor(0.0)})
    mm.update(None)
    assert str(mm) == "loss 4.0000 (3.0000) | LossM 0.0000 (2.0000)"
    with pytest.raises(TypeError):
        mm.update([1, 2])


def test_deferred_summary_resolves_once_on_first_look_and_behaves_like_a_dict():
    calls = []

    def read():
        calls.append(1)
        return {"loss": 1.5, "acc": 50.0}

    d = DeferredSummary(("loss", "acc"), read)
    assert isinstance(d, dict) and len(d) == 2 and list(d.keys()) == ["loss", "acc"] and "loss" in d
    assert not d.resolved and not calls            # nothing above needed the numbers
    assert d["loss"] == 1.5 and d.resolved and calls == [1]
    assert dict(d) == {"loss": 1.5, "acc": 50.0} and d == {"loss": 1.5, "acc": 50.0} and calls == [1]
    for make in (lambda x: dict(x), lambda x: {**x}, lambda x: x.copy(), lambda x: dict(x.items()),
                 lambda x: json.loads(json.dumps(x)), lambda x: {k: x.get(k) for k in x}):
        assert make(DeferredSummary(("loss", "acc"), read)) == {"loss": 1.5, "acc": 50.0}
    assert list(DeferredSummary(("loss", "acc"), read).values()) == [1.5, 50.0]
    assert "1.5" in repr(DeferredSummary(("loss", "acc"), read))


def test_metric_meter_folds_deferred_summaries_in_order_when_read():
    log, seen = MetricMeter(), []

    def reader(i):
        def read():
            seen.append(i)
            return {"loss": float(i)}
        return read

    for i in (1, 2, 3):
        log.update(DeferredSummary(("loss",), reader(i)))
    assert seen == []                               # no device wait while the loop only feeds the meter
    log.update({"loss": 6.0})                       # an eager dict settles what came before it, in order
    assert seen == [1, 2, 3]
    assert log.meters["loss"].val == 6.0 and log.meters["loss"].avg == 3.0 and log.meters["loss"].count == 4
    log.update(DeferredSummary(("loss",), reader(4)))
    assert str(log) == "loss 4.0000 (3.2000)" and seen == [1, 2, 3, 4]


def test_deferred_summary_pickles_and_copies_as_the_settled_dict():
    """a DeferredSummary's resolver is a closure over device state; what travels through pickle / copy is the plain dict"""
    import copy
    import pickle
    from ieee_amd.meters import DeferredSummary
    calls = []

    def resolver():
        calls.append(1)
        return {"loss": 1.5, "acc": 50.0}
    d = DeferredSummary(("loss", "acc"), resolver)
    assert not d.resolved
    back = pickle.loads(pickle.dumps(d))
    assert type(back) is dict and back == {"loss": 1.5, "acc": 50.0} and d.resolved and len(calls) == 1
    d2 = DeferredSummary(("loss", "acc"), resolver)
    deep = copy.deepcopy(d2)
    assert type(deep) is dict and deep == {"loss": 1.5, "acc": 50.0}
    d3 = DeferredSummary(("loss", "acc"), resolver)
    assert copy.copy(d3) == {"loss": 1.5, "acc": 50.0} and list(d3) == ["loss", "acc"]
