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


class MetricMeter(object):
    """one AverageMeter per key of the dicts it is fed; tensors are read with .item() (the reference's `LossM` entry is
    a 0-d tensor, engine/image/margin.py:145)"""

    def __init__(self, delimiter='\t'):
        self.meters = {}
        self.delimiter = delimiter

    def update(self, input_dict):
        if input_dict is None:
            return
        if not isinstance(input_dict, dict):
            raise TypeError('Input to MetricMeter.update() must be a dictionary')
        for key, value in input_dict.items():
            if torch.is_tensor(value):
                value = value.item()
            self.meters.setdefault(key, AverageMeter()).update(value)

    def __str__(self):
        return self.delimiter.join('%s %.4f (%.4f)' % (k, m.val, m.avg) for k, m in self.meters.items())
