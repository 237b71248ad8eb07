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
        # the update also writes the bf16 image of the new parameters (ieee_sgd_nesterov_step_shadow): the bf16 training forward
        # then reads the 1x1 convolutions' GEMM operands from that shadow instead of packing them (IEEE_SGD_SHADOW=0: off)
        import os
        model._shadow_enabled = os.environ.get("IEEE_SGD_SHADOW", "1") != "0" and getattr(model, "compute_dtype", None) == torch.bfloat16
        self._parts_done = set()

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
        self._opt_called = True                   # what torch's schedulers look at to order step() calls (step_part too)
        shadow = m.shadow_buffer() if getattr(m, "_shadow_enabled", False) else None
        # skip_flags (set by the engine for the duration of a fused step): the executor's range-guard words -- a step whose
        # BatchNorm sums were clamped leaves parameters, momentum and shadow untouched (include/ieee_amd.h)
        skip = getattr(self, "skip_flags", None)
        for a, b in runs:
            _lib.check(lib.ieee_sgd_nesterov_step_ex(_lib.ptr(m._flat_params[a:b]), _lib.ptr(m._flat_grads[a:b]),
                                                     _lib.ptr(buf[a:b]), b - a, float(g['lr']), float(g['momentum']),
                                                     float(g['weight_decay']), 1 if g['nesterov'] else 0,
                                                     _lib.ptr(shadow[a:b]) if shadow is not None else None,
                                                     _lib.ptr(skip) if skip is not None else None, _lib.stream()))

    @torch.no_grad()
    def step(self, closure=None):
        """uses the gradients the native backward left in the model's flat gradient buffer"""
        self._update(self.model.trainable_runs())
        self.model.shadow_is_current()

    @torch.no_grad()
    def step_part(self, part):
        """the same update restricted to the parameters whose gradients are final after staged-backward part `part`
        (model.part_runs()); the five parts together are exactly step().  Runs on the current stream."""
        parts = self.model.part_runs()
        self._update(parts[part])
        if part == 0:
            self._parts_done = set()
        self._parts_done.add(part)
        if len(self._parts_done) == len(parts):   # every trainable parameter (and its shadow element) has been written
            self._parts_done = set()
            self.model.shadow_is_current()

    def zero_grad(self, set_to_none=True):
        for p in self.model.parameters():
            p.grad = None

    def flat_state(self):
        """the flat state tensors (data-parallel replica synchronisation)"""
        return [self.momentum_buffer()]

    # Checkpoint interop with torch.optim.SGD (the reference's optimizer, optim/optimizer.py:130-138): state_dict()
    # carries the momentum as per-parameter `momentum_buffer` entries -- views of the flat buffer, so the file stores
    # it once -- exactly where torch.optim.SGD keeps its own, and load_state_dict() scatters such entries (from either
    # implementation) back into the flat buffer.
    def _publish_views(self):
        buf = self.momentum_buffer()
        for name, p in self.model._param_items:
            off = self.model._offsets[name]
            self.state[p]['momentum_buffer'] = buf[off:off + p.numel()].view(p.shape)

    def state_dict(self):
        self._publish_views()
        d = super(FusedSGD, self).state_dict()
        d['fused'] = {'layout': 'per-parameter views of one flat buffer'}
        return d

    def load_state_dict(self, state_dict):
        fused = state_dict.get('fused')
        super(FusedSGD, self).load_state_dict({k: v for k, v in state_dict.items() if k != 'fused'})
        buf = self.momentum_buffer()
        if fused is not None and fused.get('momentum_buffer') is not None:        # files written by round-1 builds
            buf.copy_(fused['momentum_buffer'])
        else:
            buf.zero_()
            for name, p in self.model._param_items:
                mb = self.state.get(p, {}).get('momentum_buffer')
                if mb is not None:
                    off = self.model._offsets[name]
                    buf[off:off + p.numel()].copy_(mb.reshape(-1))
        self._publish_views()


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam(lr, betas, eps=1e-8, weight_decay[, amsgrad]) over the model's flat buffers in one launch per
    contiguous trainable run (reference optim/optimizer.py:113-128 builds exactly these two variants)."""

    def __init__(self, model, lr=0.0003, betas=(0.9, 0.99), eps=1e-8, weight_decay=5e-4, amsgrad=False):
        self.model = model
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad)
        super(FusedAdam, self).__init__(list(model.parameters()), defaults)
        model._shadow_enabled = False              # (the bf16 parameter shadow is FusedSGD's: every operand is packed here)
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

    def flat_state(self):
        return [t for t in self._buffers() if t is not None]

    # same checkpoint interop as FusedSGD, with torch.optim.Adam's per-parameter keys (step, exp_avg, exp_avg_sq,
    # max_exp_avg_sq)
    _KEYS = ('exp_avg', 'exp_avg_sq', 'max_exp_avg_sq')

    def _publish_views(self):
        flats = self._buffers()
        for name, p in self.model._param_items:
            off, n = self.model._offsets[name], p.numel()
            st = self.state[p]
            st['step'] = torch.tensor(float(self._step))
            for key, flat in zip(self._KEYS, flats):
                if flat is not None:
                    st[key] = flat[off:off + n].view(p.shape)

    def state_dict(self):
        self._publish_views()
        d = super(FusedAdam, self).state_dict()
        d['fused'] = {'layout': 'per-parameter views of flat buffers', 'step': self._step}
        return d

    def load_state_dict(self, state_dict):
        fused = state_dict.get('fused')
        super(FusedAdam, self).load_state_dict({k: v for k, v in state_dict.items() if k != 'fused'})
        flats = self._buffers()
        if fused is not None and fused.get('exp_avg') is not None:                  # files written by round-1 builds
            self._step = int(fused['step'])
            flats[0].copy_(fused['exp_avg']); flats[1].copy_(fused['exp_avg_sq'])
            if flats[2] is not None and fused.get('max_exp_avg_sq') is not None:
                flats[2].copy_(fused['max_exp_avg_sq'])
        else:
            steps = []
            for flat in flats:
                if flat is not None:
                    flat.zero_()
            for name, p in self.model._param_items:
                st = self.state.get(p, {})
                off, n = self.model._offsets[name], p.numel()
                if 'step' in st:
                    steps.append(int(float(st['step'])))
                for key, flat in zip(self._KEYS, flats):
                    if flat is not None and st.get(key) is not None:
                        flat[off:off + n].copy_(st[key].reshape(-1))
            self._step = max(steps) if steps else (int(fused['step']) if fused and 'step' in fused else 0)
        self._publish_views()


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
