"""Mirror of torchreid.optim for the configuration on the hot path (reference
torchreid/optim/optimizer.py:11-157, lr_scheduler.py:7-68): SGD(momentum, weight_decay, dampening=0,
nesterov=True) with a MultiStepLR.  FusedSGD runs ieee_sgd_nesterov_step over the model's flat
parameter / gradient buffers (one launch per contiguous trainable run) and is a torch.optim.Optimizer,
so schedulers, state_dict() and the reference's checkpoint code keep working."""
import torch

from . import _lib

AVAI_OPTIMS = ['adam', 'amsgrad', 'sgd', 'rmsprop', 'radam']
AVAI_SCH = ['single_step', 'multi_step', 'cosine']


class FusedSGD(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-3, momentum=0.9, weight_decay=5e-4, nesterov=True):
        self.model = model
        defaults = dict(lr=lr, momentum=momentum, weight_decay=weight_decay, nesterov=nesterov, dampening=0)
        super(FusedSGD, self).__init__(list(model.parameters()), defaults)
        self._buf = None

    def momentum_buffer(self):
        if self._buf is None or self._buf.device != self.model._flat_params.device:
            self._buf = torch.zeros_like(self.model._flat_params)
        return self._buf

    def _update(self, runs):
        lib = _lib.require_gpu()
        g = self.param_groups[0]
        m = self.model
        buf = self.momentum_buffer()
        m._native_epoch += 1                      # parameters change behind torch's version counters
        for a, b in runs:
            _lib.check(lib.ieee_sgd_nesterov_step(_lib.ptr(m._flat_params[a:b]), _lib.ptr(m._flat_grads[a:b]),
                                                  _lib.ptr(buf[a:b]), b - a, float(g['lr']), float(g['momentum']),
                                                  float(g['weight_decay']), 1 if g['nesterov'] else 0,
                                                  _lib.stream()))

    @torch.no_grad()
    def step(self, closure=None):
        """uses the gradients the native backward left in the model's flat gradient buffer"""
        self._update(self.model.trainable_runs())

    @torch.no_grad()
    def step_part(self, part):
        """the same update restricted to the parameters whose gradients are final after staged-backward part `part`
        (model.part_runs()); the five parts together are exactly step().  Runs on the current stream."""
        self._update(self.model.part_runs()[part])

    def zero_grad(self, set_to_none=True):
        for p in self.model.parameters():
            p.grad = None

    def state_dict(self):
        d = super(FusedSGD, self).state_dict()
        d['fused'] = {'momentum_buffer': self._buf}
        return d

    def load_state_dict(self, state_dict):
        fused = state_dict.get('fused')
        super(FusedSGD, self).load_state_dict({k: v for k, v in state_dict.items() if k != 'fused'})
        if fused is not None and fused.get('momentum_buffer') is not None:
            self.momentum_buffer().copy_(fused['momentum_buffer'])


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam(lr, betas, eps=1e-8, weight_decay[, amsgrad]) over the model's flat buffers in one launch per
    contiguous trainable run (reference optim/optimizer.py:113-128 builds exactly these two variants)."""

    def __init__(self, model, lr=0.0003, betas=(0.9, 0.99), eps=1e-8, weight_decay=5e-4, amsgrad=False):
        self.model = model
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad)
        super(FusedAdam, self).__init__(list(model.parameters()), defaults)
        self._m = self._v = self._vmax = None
        self._step = 0

    def _buffers(self):
        ref = self.model._flat_params
        if self._m is None or self._m.device != ref.device:
            self._m, self._v = torch.zeros_like(ref), torch.zeros_like(ref)
            self._vmax = torch.zeros_like(ref) if self.param_groups[0]['amsgrad'] else None
        return self._m, self._v, self._vmax

    @torch.no_grad()
    def step(self, closure=None):
        lib = _lib.require_gpu()
        g = self.param_groups[0]
        mdl = self.model
        m, v, vmax = self._buffers()
        self._step += 1
        mdl._native_epoch += 1                    # parameters change behind torch's version counters
        for a, b in mdl.trainable_runs():
            _lib.check(lib.ieee_adam_step(_lib.ptr(mdl._flat_params[a:b]), _lib.ptr(mdl._flat_grads[a:b]), _lib.ptr(m[a:b]),
                                          _lib.ptr(v[a:b]), _lib.ptr(vmax[a:b]) if vmax is not None else None, b - a,
                                          float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']),
                                          float(g['weight_decay']), self._step, _lib.stream()))

    def zero_grad(self, set_to_none=True):
        for p in self.model.parameters():
            p.grad = None

    def state_dict(self):
        d = super(FusedAdam, self).state_dict()
        d['fused'] = {'step': self._step, 'exp_avg': self._m, 'exp_avg_sq': self._v, 'max_exp_avg_sq': self._vmax}
        return d

    def load_state_dict(self, state_dict):
        fused = state_dict.get('fused')
        super(FusedAdam, self).load_state_dict({k: v for k, v in state_dict.items() if k != 'fused'})
        if fused is not None:
            m, v, vmax = self._buffers()
            self._step = int(fused['step'])
            if fused['exp_avg'] is not None:
                m.copy_(fused['exp_avg']); v.copy_(fused['exp_avg_sq'])
                if vmax is not None and fused['max_exp_avg_sq'] is not None:
                    vmax.copy_(fused['max_exp_avg_sq'])


def build_optimizer(model, optim='adam', lr=0.0003, weight_decay=5e-04, momentum=0.9, sgd_dampening=0,
                    sgd_nesterov=False, rmsprop_alpha=0.99, adam_beta1=0.9, adam_beta2=0.99, staged_lr=False,
                    new_layers='', base_lr_mult=0.1, fused=True):
    """reference optim/optimizer.py:11-157.  Note the reference's SGD branch hard-codes nesterov=True
    (:137) whatever `sgd_nesterov` says; kept.  `fused=True` (SGD only) returns FusedSGD."""
    if optim not in AVAI_OPTIMS:
        raise ValueError('Unsupported optim: {}. Must be one of {}'.format(optim, AVAI_OPTIMS))
    if not isinstance(model, torch.nn.Module):
        raise TypeError('model given to build_optimizer must be an instance of nn.Module')
    if staged_lr:
        raise NotImplementedError("staged_lr is off in the reference's config (default_config.py:58)")
    if isinstance(model, torch.nn.DataParallel):
        model = model.module
    params = model.parameters()
    if optim == 'sgd':
        if fused and hasattr(model, "trainable_runs"):
            return FusedSGD(model, lr=lr, momentum=momentum, weight_decay=weight_decay, nesterov=True)
        return torch.optim.SGD(params, lr=lr, momentum=momentum, weight_decay=weight_decay,
                               dampening=sgd_dampening, nesterov=True)
    if optim in ('adam', 'amsgrad') and fused and hasattr(model, "trainable_runs"):
        return FusedAdam(model, lr=lr, betas=(adam_beta1, adam_beta2), weight_decay=weight_decay,
                         amsgrad=(optim == 'amsgrad'))
    if optim == 'adam':
        return torch.optim.Adam(params, lr=lr, weight_decay=weight_decay, betas=(adam_beta1, adam_beta2))
    if optim == 'amsgrad':
        return torch.optim.Adam(params, lr=lr, weight_decay=weight_decay, betas=(adam_beta1, adam_beta2),
                                amsgrad=True)
    if optim == 'rmsprop':
        return torch.optim.RMSprop(params, lr=lr, momentum=momentum, weight_decay=weight_decay, alpha=rmsprop_alpha)
    return torch.optim.RAdam(params, lr=lr, weight_decay=weight_decay, betas=(adam_beta1, adam_beta2))


def build_lr_scheduler(optimizer, lr_scheduler='single_step', stepsize=1, gamma=0.1, max_epoch=1):
    """reference optim/lr_scheduler.py:7-68"""
    if lr_scheduler not in AVAI_SCH:
        raise ValueError('Unsupported scheduler: {}. Must be one of {}'.format(lr_scheduler, AVAI_SCH))
    if lr_scheduler == 'single_step':
        if isinstance(stepsize, list):
            stepsize = stepsize[-1]
        if not isinstance(stepsize, int):
            raise TypeError('For single_step lr_scheduler, stepsize must be an integer, but got {}'.format(
                type(stepsize)))
        return torch.optim.lr_scheduler.StepLR(optimizer, step_size=stepsize, gamma=gamma)
    if lr_scheduler == 'multi_step':
        if not isinstance(stepsize, list):
            raise TypeError('For multi_step lr_scheduler, stepsize must be a list, but got {}'.format(type(stepsize)))
        return torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=stepsize, gamma=gamma)
    return torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, float(max_epoch))
