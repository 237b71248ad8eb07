"""top-k accuracy in percent, the logging-only output of the train step (reference torchreid/metrics/accuracy.py:4-38;
SURVEY.md §8a A21): one [1]-shaped tensor per requested k, on whatever device the logits live on."""


def accuracy(output, target, topk=(1, )):
    logits = output[0] if isinstance(output, (tuple, list)) else output
    n = target.size(0)
    ranked = logits.topk(max(topk), dim=1, largest=True, sorted=True).indices          # [n, max k] class ids, best first
    hit = ranked.eq(target.reshape(-1, 1))                                             # a label matches at most one column
    return [hit[:, :k].any(dim=1).float().sum().reshape(1) * (100.0 / n) for k in topk]
