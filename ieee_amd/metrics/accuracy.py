"""top-k accuracy, reference torchreid/metrics/accuracy.py:4-38 (logging-only output of the train
step).  Plain torch indexing on whatever device the logits live on."""


def accuracy(output, target, topk=(1, )):
    maxk = max(topk)
    batch_size = target.size(0)
    if isinstance(output, (tuple, list)):
        output = output[0]
    _, pred = output.topk(maxk, 1, True, True)
    pred = pred.t()
    correct = pred.eq(target.view(1, -1).expand_as(pred))
    res = []
    for k in topk:
        correct_k = correct[:k].reshape(-1).float().sum(0, keepdim=True)
        res.append(correct_k.mul_(100.0 / batch_size))
    return res
