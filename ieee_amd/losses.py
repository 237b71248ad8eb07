"""Mirror of torchreid.losses for the two criteria on the hot path.  Same classes / call signatures as
the reference (torchreid/losses/cross_entropy_loss.py:6-50, multi_modal_margin_loss_new.py:7-40,
losses/__init__.py:9-44); forward AND backward run in ieee_ce_ls_fwd_bwd / ieee_margin3m_fwd_bwd."""
import torch
from torch import nn

from . import _lib


class _CEFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets, eps):
        lib = _lib.require_gpu()
        x = logits.to(torch.float32).contiguous()
        squeeze = x.dim() == 2
        if squeeze:
            x = x.unsqueeze(0)
        heads, B, C = x.shape
        t = targets.to(device=x.device, dtype=torch.int64).contiguous()
        dl = torch.empty_like(x)
        head_loss = torch.empty(heads, dtype=torch.float32, device=x.device)
        head_acc = torch.empty(heads, dtype=torch.float32, device=x.device)
        work = torch.empty(heads * B * 2, dtype=torch.float32, device=x.device)
        _lib.check(lib.ieee_ce_ls_fwd_bwd(_lib.ptr(x), _lib.ptr(t), _lib.ptr(dl), _lib.ptr(head_loss),
                                          _lib.ptr(head_acc), _lib.ptr(work), heads, B, C, float(eps), 1.0,
                                          _lib.stream()))
        ctx.save_for_backward(dl)
        ctx.squeeze = squeeze
        return head_loss[0] if squeeze else head_loss


    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        if ctx.squeeze:
            return dl[0] * g, None, None
        return dl * g.view(-1, 1, 1), None, None


class CrossEntropyLoss(nn.Module):
    r"""Cross entropy loss with label smoothing regularizer (reference cross_entropy_loss.py:6-50):
    ``(-t * log_softmax(x)).mean(0).sum()`` with ``t = (1-eps)*onehot + eps/K``.  The reference builds
    the one-hot on the host (`targets.cpu()`, :46); here nothing leaves the device."""

    def __init__(self, num_classes, eps=0.1, use_gpu=True, label_smooth=True):
        super(CrossEntropyLoss, self).__init__()
        self.num_classes = num_classes
        self.eps = eps if label_smooth else 0
        self.use_gpu = use_gpu

    def forward(self, inputs, targets):
        assert inputs.dim() == 2 and inputs.size(1) == self.num_classes
        return _CEFunction.apply(inputs, targets, self.eps)


class _MarginFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f1, f2, f3, pids, margin):
        lib = _lib.require_gpu()
        feats = torch.stack([f1, f2, f3]).to(torch.float32).contiguous()
        _, B, D = feats.shape
        p = pids.to(device=feats.device, dtype=torch.int64).contiguous()
        df = torch.empty_like(feats)
        out3 = torch.empty(3, dtype=torch.float32, device=feats.device)
        work = torch.empty(B + 3, dtype=torch.float32, device=feats.device)
        _lib.check(lib.ieee_margin3m_fwd_bwd(_lib.ptr(feats), _lib.ptr(p), _lib.ptr(df), _lib.ptr(out3), _lib.ptr(work), B, D,
                                             float(margin), 1.0, _lib.stream()))
        ctx.save_for_backward(df)
        ctx.out3 = out3
        return out3[0]

    @staticmethod
    def backward(ctx, g):
        (df,) = ctx.saved_tensors
        return df[0] * g, df[1] * g, df[2] * g, None, None


class multiModalMarginLossNew(nn.Module):
    """3M loss, reference multi_modal_margin_loss_new.py:7-40 (dist_type 'l2' is the one the engine uses:
    margin.py:80-91 constructs it with the default).  Per identity chunk: centers per modality,
    max(|m-d(1,2)|, |m-d(2,3)|, |m-d(1,3)|) with d = squared L2, summed over chunks."""

    def __init__(self, margin=3, dist_type='l2'):
        super(multiModalMarginLossNew, self).__init__()
        if dist_type != 'l2':
            raise NotImplementedError("only dist_type='l2' is on the hot path (SURVEY.md §8a A13)")
        self.dist_type = dist_type
        self.margin = margin

    def forward(self, feat1, feat2, feat3, label1):
        return _MarginFunction.apply(feat1, feat2, feat3, label1, self.margin)


def DeepSupervision(criterion, xs, y):
    """reference losses/__init__.py:9-44: sum of the criterion over a list of outputs.  When the list is the
    native model's 6 part-logits (views of one [18,B,C] tensor) this is still 6 small launches; the fused
    engine path batches all 18 heads in one."""
    loss = 0.
    for x in xs:
        loss += criterion(x, y)
    return loss
