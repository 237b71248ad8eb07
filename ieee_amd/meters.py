"""Running averages for the training log (what the reference keeps in utils/avgmeter.py:8-73)."""
import torch


class AverageMeter(object):
    """last value, running sum, count and mean of one scalar"""

    __slots__ = ("val", "sum", "count")

    def __init__(self):
        self.reset()

    def reset(self):
        self.val, self.sum, self.count = 0, 0, 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n

    @property
    def avg(self):
        return self.sum / self.count if self.count else 0


class DeferredSummary(dict):
    """A loss_summary whose numbers are still on their way from the device: the step enqueued its read-back (into a
    pinned buffer, on the step's own stream) and returned; the first look at any value waits for that copy and fills the
    dict.  Keys and their order are known from the start.  The reference reads `loss.item()` inside every
    forward_backward (engine/image/margin.py:143-152), i.e. drains the device once per step; a training loop that only
    prints every print_freq batches does not need to (`Engine.defer_summary`, MetricMeter below)."""

    def __init__(self, keys, resolver):
        super(DeferredSummary, self).__init__((k, None) for k in keys)
        self._resolver = resolver

    @property
    def resolved(self):
        return self._resolver is None

    def resolve(self):
        if self._resolver is not None:
            resolver, self._resolver = self._resolver, None
            super(DeferredSummary, self).update(resolver())
        return self

    def __getitem__(self, key):
        self.resolve()
        return super(DeferredSummary, self).__getitem__(key)

    def get(self, key, default=None):
        self.resolve()
        return super(DeferredSummary, self).get(key, default)

    def __iter__(self):                  # also keeps dict(x) / {**x} off the C fast path that would copy the placeholders
        self.resolve()
        return super(DeferredSummary, self).__iter__()

    def items(self):
        self.resolve()
        return super(DeferredSummary, self).items()

    def values(self):
        self.resolve()
        return super(DeferredSummary, self).values()

    def copy(self):
        self.resolve()
        return dict(super(DeferredSummary, self).items())

    def __eq__(self, other):
        self.resolve()
        return super(DeferredSummary, self).__eq__(other)

    __hash__ = None

    # pickling / copy.deepcopy: the resolver is a closure over device state; what travels is the settled plain dict
    def __reduce__(self):
        self.resolve()
        return (dict, (dict(super(DeferredSummary, self).items()),))

    def __deepcopy__(self, memo):
        import copy
        self.resolve()
        return copy.deepcopy(dict(super(DeferredSummary, self).items()), memo)

    def __copy__(self):
        return self.copy()

    def __repr__(self):
        self.resolve()
        return super(DeferredSummary, self).__repr__()


class MetricMeter(object):
    """one AverageMeter per key of the dicts it is fed; tensors are read with .item() (the reference's `LossM` entry is
    a 0-d tensor, engine/image/margin.py:145)"""

    def __init__(self, delimiter='\t'):
        self._meters = {}
        self._pending = []               # DeferredSummary objects not looked at yet, oldest first
        self.delimiter = delimiter

    def update(self, input_dict):
        if input_dict is None:
            return
        if not isinstance(input_dict, dict):
            raise TypeError('Input to MetricMeter.update() must be a dictionary')
        if isinstance(input_dict, DeferredSummary) and not input_dict.resolved:
            self._pending.append(input_dict)     # folded in, in order, when somebody reads the meters
            return
        self._flush()
        self._fold(input_dict)

    def _fold(self, d):
        for key, value in d.items():
            if torch.is_tensor(value):
                value = value.item()
            self._meters.setdefault(key, AverageMeter()).update(value)

    def _flush(self):
        pending, self._pending = self._pending, []
        for d in pending:
            self._fold(d)

    @property
    def meters(self):
        self._flush()
        return self._meters

    def __str__(self):
        return self.delimiter.join('%s %.4f (%.4f)' % (k, m.val, m.avg) for k, m in self.meters.items())
