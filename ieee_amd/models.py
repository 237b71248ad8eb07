"""Host-side mirror of torchreid.models for the one model on the hot path.

`build_model('ieee3modalPart', ...)` and `IEEE3modalPart` keep the reference's constructor, forward
contract, parameter names/order and state_dict keys (torchreid/models/__init__.py:80-111,
torchreid/models/ieee3modalPart.py:286-555), so `scripts/mainMultiModal.py`-style callers, torch.optim
optimizers and the reference's checkpoint helpers work unchanged.  The arithmetic does not live here:
forward/backward are one call each into the native executor (ieee_net_forward / ieee_net_backward),
which runs the hand-written HIP kernels.  No CPU fallback."""
from __future__ import absolute_import, division

import math

import torch
from torch import nn

from . import _lib
from ._net import NativeNet
from ._spec import state_spec

__all__ = ['build_model', 'show_avai_models', 'ieee3modalPart', 'IEEE3modalPart']


class _Node(nn.Module):
    """plain container: the parameter tree only carries names (state_dict keys), not compute"""

    def forward(self, *a, **k):
        raise RuntimeError("this module only holds parameters; call IEEE3modalPart.forward")

    # the reference's containers are ModuleList / Sequential: keep `model.backbone[0].layer4[2].conv3.weight` working
    def __getitem__(self, i):
        return self._modules[str(i)]

    def __len__(self):
        return len(self._modules)


class _NetFunction(torch.autograd.Function):
    """autograd bridge for the drop-in path (any torch.optim over model.parameters())"""

    @staticmethod
    def forward(ctx, model, net, xr, xn, xt, *params):
        logits, feats = net.forward([xr, xn, xt], training=True)
        ctx.model, ctx.net = model, net
        return logits, feats

    @staticmethod
    def backward(ctx, dlogits, dfeats):
        model, net = ctx.model, ctx.net
        net.backward(dlogits, dfeats)
        grads = []
        for name, p in model._param_items:
            if name in model._no_grad_names():
                grads.append(None)
            else:
                off = model._offsets[name]
                grads.append(model._flat_grads[off:off + p.numel()].view(p.shape).clone())
        return (None, None, None, None, None) + tuple(grads)


class IEEE3modalPart(nn.Module):
    """Same signature as the reference class (ieee3modalPart.py:286-297).  The three ablation switches
    the reference hard-codes as attributes (:312-314) are constructor keywords here (default True)."""

    def __init__(self, num_classes, loss, block=None, parts=1, reduced_dim=512, cls_dim=128, nonlinear='relu',
                 pretrained=True, interaction=True, attention=True, using_REM=True, compute_dtype=None,
                 device=None, **kwargs):
        super(IEEE3modalPart, self).__init__()
        self.loss = loss
        self.parts = 6
        self.num_classes = num_classes
        self.interaction = interaction
        self.attention = attention
        self.using_REM = using_REM
        if compute_dtype is None:
            # bf16 storage / fp32 accumulate is the speed mode BASELINE config 2 names; the fp32 PARITY mode (exact fp32
            # MFMA end to end: logits / features within 1e-3 of the reference's CPU path) is compute_dtype=torch.float32
            # or IEEE_COMPUTE_DTYPE=fp32 for unchanged callers
            import os
            if "IEEE_COMPUTE_DTYPE" not in os.environ:
                # a caller who only swapped the import leaves the reference's fp32 arithmetic without having said so: say it once
                import warnings
                warnings.warn("ieee_amd: IEEE3modalPart built without compute_dtype runs in the bf16 SPEED mode (bf16 storage, fp32 "
                              "accumulate: BASELINE config 2); the reference's fp32 arithmetic -- logits / features within 1e-3 -- is "
                              "the PARITY mode: build_model(..., compute_dtype=torch.float32) or IEEE_COMPUTE_DTYPE=fp32.  Pass "
                              "compute_dtype=torch.bfloat16 (or set IEEE_COMPUTE_DTYPE=bf16) to choose the speed mode silently.",
                              RuntimeWarning, stacklevel=3)
            env = os.environ.get("IEEE_COMPUTE_DTYPE", "bf16").lower()
            if env not in ("bf16", "fp32", "float32", "bfloat16"):
                raise ValueError("IEEE_COMPUTE_DTYPE must be bf16 or fp32, got %r" % env)
            compute_dtype = torch.float32 if env in ("fp32", "float32") else torch.bfloat16
        self.compute_dtype = compute_dtype
        self._nets = {}
        self._native_epoch = 0                   # bumped by every native writer of parameters / running statistics
        self._spec = state_spec(num_classes)     # all children exist (keys identical to the default reference)
        self._build_storage(device)
        self._init_params()
        if pretrained:
            # reference: each backbone downloads resnet50-19c8e357.pth (resnet.py:25-26, 1259-1261).  No network here:
            # use the file if it is already on disk ($IEEE_RESNET50_PTH or the torch hub cache), else say so.
            from . import checkpoint as _ckpt
            path = _ckpt.find_resnet50_file()
            if path is None:
                raise RuntimeError(
                    "pretrained=True needs %s (reference resnet.py:25-26, 1075-1089) and there is no network: put the "
                    "file in the torch hub cache or set IEEE_RESNET50_PTH, or build with pretrained=False and call "
                    "ieee_amd.checkpoint.init_pretrained_backbones / load_pretrained_weights." % _ckpt.RESNET50_FILE)
            print("load pretrained weights...")
            _ckpt.init_pretrained_backbones(self, path)

    # ---- storage: one flat fp32 buffer per kind, parameters are views in state_dict order
    def _build_storage(self, device):
        if device is None:
            device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
        n_param = sum(int(math.prod(s)) for _, s, k in self._spec if k == "param")
        n_buf = sum(int(math.prod(s)) for _, s, k in self._spec if k == "buffer")
        n_cnt = sum(1 for _, _, k in self._spec if k == "counter")
        self._flat_params = torch.zeros(n_param, dtype=torch.float32, device=device)
        self._flat_grads = torch.zeros(n_param, dtype=torch.float32, device=device)
        self._flat_buffers = torch.zeros(n_buf, dtype=torch.float32, device=device)
        self._flat_counters = torch.zeros(n_cnt, dtype=torch.int64, device=device)
        self._offsets = {}
        self._param_items = []
        po = bo = co = 0
        for key, shape, kind in self._spec:
            node = self
            parts = key.split(".")
            for name in parts[:-1]:
                if name not in node._modules:
                    node.add_module(name, _Node())
                node = node._modules[name]
            n = int(math.prod(shape))
            if kind == "param":
                p = nn.Parameter(self._flat_params[po:po + n].view(shape))
                node.register_parameter(parts[-1], p)
                self._offsets[key] = po
                self._param_items.append((key, p))
                po += n
            elif kind == "buffer":
                node.register_buffer(parts[-1], self._flat_buffers[bo:bo + n].view(shape))
                self._offsets[key] = bo
                bo += n
            else:
                node.register_buffer(parts[-1], self._flat_counters[co:co + 1].view(shape))
                self._offsets[key] = co
                co += 1

    def _init_params(self):
        """trunks: kaiming_normal(fan_out)/BN(1,0) (resnet.py:603-620); everything else torch's defaults,
        REM.param = 0 (ieee3modalPart.py:58)."""
        with torch.no_grad():
            params = dict(self.named_parameters())
            bufs = dict(self.named_buffers())
            for key, shape, kind in self._spec:
                leaf = key.rsplit(".", 1)[-1]
                if kind == "counter":
                    continue
                if kind == "buffer":
                    bufs[key].fill_(1.0 if leaf == "running_var" else 0.0)
                    continue
                p = params[key]
                if leaf == "param":
                    p.zero_()
                elif len(shape) == 4:
                    if key.startswith("backbone."):
                        nn.init.kaiming_normal_(p, mode='fan_out', nonlinearity='relu')
                    else:
                        nn.init.kaiming_uniform_(p, a=math.sqrt(5))
                elif len(shape) == 2:
                    nn.init.kaiming_uniform_(p, a=math.sqrt(5))
                elif leaf == "bias" and key.replace(".bias", ".weight") in params and \
                        params[key.replace(".bias", ".weight")].dim() == 2:
                    fan_in = params[key.replace(".bias", ".weight")].shape[1]
                    bound = 1.0 / math.sqrt(fan_in)
                    nn.init.uniform_(p, -bound, bound)
                elif leaf == "weight":
                    p.fill_(1.0)
                else:
                    p.zero_()

    def _apply(self, fn, *args, **kwargs):
        """device moves re-create the flat buffers and the parameter views"""
        probe = fn(torch.zeros(1, dtype=torch.float32, device=self._flat_params.device))
        if probe.device != self._flat_params.device or probe.dtype != torch.float32:
            if probe.dtype != torch.float32:
                raise RuntimeError("IEEE3modalPart keeps fp32 master parameters; use compute_dtype for bf16 math")
            state = {k: v.detach().clone() for k, v in self.state_dict().items()}
            self._modules.clear()
            self._nets = {}
            self._build_storage(probe.device)
            self.load_state_dict(state)
        return self

    def _no_grad_names(self):
        """parameters that receive no gradient (grad None in the reference): REM.conv_value always
        (SURVEY.md §8a A7); whole branches when their ablation flag is off."""
        if getattr(self, "_no_grad_cache_key", None) != (self.interaction, self.attention, self.using_REM):
            s = set()
            for key, p in self._param_items:
                if ".conv_value." in key:
                    s.add(key)
                if not self.using_REM and key.startswith("REM."):
                    s.add(key)
                if not self.interaction and key.split(".")[0] in ("convOne", "convAvgRest", "CA"):
                    s.add(key)
                if not self.attention and key.startswith("CA."):
                    s.add(key)
            self._no_grad_cache = s
            self._no_grad_cache_key = (self.interaction, self.attention, self.using_REM)
        return self._no_grad_cache

    def trainable_runs(self):
        """contiguous [start, end) element runs of the flat parameter buffer that receive gradients
        (what the fused SGD step iterates over)."""
        skip = self._no_grad_names()
        runs, start, pos = [], None, 0
        for key, p in self._param_items:
            n = p.numel()
            if key in skip or not p.requires_grad:
                if start is not None:
                    runs.append((start, pos))
                    start = None
            elif start is None:
                start = pos
            pos += n
        if start is not None:
            runs.append((start, pos))
        return runs

    def part_runs(self):
        """trainable_runs() cut at the boundaries of the 5 staged-backward parts: part_runs()[p] = the element runs an
        optimizer may update as soon as part p of the backward (and its weight gradients) is done"""
        # the runs depend on the ablation flags AND on which parameters are frozen (requires_grad, toggled by
        # Engine.two_stepped_transfer_learning / open_specified_layers between epochs)
        key = (self.interaction, self.attention, self.using_REM, tuple(p.requires_grad for _, p in self._param_items))
        if getattr(self, "_part_runs_key", None) != key:
            runs = self.trainable_runs()
            out = []
            for ranges in self.grad_part_ranges():
                mine = []
                for a, b in ranges:
                    for c, d in runs:
                        lo, hi = max(a, c), min(b, d)
                        if lo < hi:
                            mine.append((lo, hi))
                out.append(mine)
            self._part_runs, self._part_runs_key = out, key
        return self._part_runs

    _FROZEN_BITS = {"backbone": 1, "convOne": 2, "convAvgRest": 4, "reduce_layer": 8, "fc_R": 16, "fc_N": 32, "fc_T": 64}

    def set_frozen_children(self, names):
        """The children that `open_specified_layers` puts in eval() mode (reference utils/torchtools.py:183-221: every child
        outside `open_layers` during the first fixbase_epoch epochs): their BatchNorms use the running statistics in a
        training forward and leave them (and num_batches_tracked) alone; the backward sees fixed affine maps
        (include/ieee_amd.h: ieee_net_set_frozen).  names = None / empty: nothing is frozen (open_all_layers).  Whether a
        frozen parameter is updated is the optimizer's business (requires_grad, Engine.two_stepped_transfer_learning)."""
        names = set(names or ())
        unknown = names - set(n for n, _ in self.named_children())
        assert not unknown, "not children of the model: %s" % sorted(unknown)
        self._frozen_children = names
        self._frozen_mask = sum(bit for n, bit in self._FROZEN_BITS.items() if n in names)
        for net in self._nets.values():
            net.set_frozen(self._frozen_mask)
        if hasattr(self, "_counter_inc"):
            del self._counter_inc                # rebuilt with the frozen children's counters standing still

    def invalidate_eval_cache(self):
        """call after writing parameters or running statistics in a way torch's version counters do not see
        (e.g. through `.data`): the next eval forward re-packs the weights and re-derives the BatchNorm scale / shift"""
        self._native_epoch += 1

    def grad_part_ranges(self):
        """[start, end) element ranges of the flat gradient buffer that are final after each of the 5 staged
        backward parts (0: head + CIM, 1: layer4, 2: layer3, 3: layer2, 4: layer1 + stem; three ranges per
        trunk part, one per modality)"""
        if not hasattr(self, "_grad_parts"):
            def part_of(key):
                if not key.startswith("backbone."):
                    return 0
                sub = key.split(".")[2]
                return {"layer4": 1, "layer3": 2, "layer2": 3}.get(sub, 4)
            parts = [[] for _ in range(5)]
            pos, cur, start = 0, None, 0
            for key, p in self._param_items:
                pid = part_of(key)
                if pid != cur:
                    if cur is not None:
                        parts[cur].append((start, pos))
                    cur, start = pid, pos
                pos += p.numel()
            parts[cur].append((start, pos))
            self._grad_parts = parts
        return self._grad_parts

    # ---- bf16 shadow of the flat parameter buffer (FusedSGD keeps it current; NativeNet hands it to the executor) ----
    _flat_shadow = None
    _shadow_key = None
    _shadow_enabled = False

    def shadow_buffer(self):
        """bf16 [numel of _flat_params]: element i = bf16(_flat_params[i]) whenever `_shadow_key` equals the parameters' current
        (torch version counter, native writer count)"""
        if self._flat_shadow is None or self._flat_shadow.device != self._flat_params.device:
            self._flat_shadow = torch.empty(self._flat_params.numel(), dtype=torch.bfloat16, device=self._flat_params.device)
            self._shadow_key = None
        return self._flat_shadow

    def shadow_is_current(self):
        """called by the optimizer when it has just written every trainable parameter together with its shadow element"""
        if self._shadow_enabled and self._flat_shadow is not None and self._shadow_key is not None:
            self._shadow_key = (self._flat_params._version, self._native_epoch)

    def fresh_shadow(self):
        """the shadow, brought up to date if anything but the shadow-writing optimizer has touched the parameters since (a
        load_state_dict, a replica sync, an in-place edit: torch's version counter; another native writer: _native_epoch)"""
        sh = self.shadow_buffer()
        key = (self._flat_params._version, self._native_epoch)
        if self._shadow_key != key:
            with torch.no_grad():
                sh.copy_(self._flat_params)          # round-to-nearest-even, as the packing kernels convert
            self._shadow_key = key
        return sh

    def native_net(self, batch, height, width):
        _lib.require_gpu()
        if self._flat_params.device.type != "cuda":
            raise _lib.IeeeAmdError("IEEE3modalPart parameters are on %s; move the model to the GPU (.cuda()): "
                                    "there is no CPU execution path." % self._flat_params.device)
        key = (batch, height, width, self.compute_dtype, self.interaction, self.attention, self.using_REM)
        net = self._nets.get(key)
        if net is None:
            self._nets.clear()       # one live workspace at a time
            net = NativeNet(self, batch, height, width, self.compute_dtype)
            net.set_frozen(getattr(self, "_frozen_mask", 0))
            if getattr(self, "_bn_totals_off", False):      # the range guard degraded this model (engine._check_bn_range)
                net.set_bn_totals(False)
            self._nets[key] = net
        return net

    def _bump_counters(self):
        """num_batches_tracked += 1 per BN forward in train mode (reduce_layer BNs run twice, :449-455)"""
        if not hasattr(self, "_counter_inc"):
            inc = []
            for key, shape, kind in self._spec:
                if kind == "counter":
                    inc.append(2 if key.startswith("reduce_layer.") else 1)
            self._counter_inc = torch.tensor(inc, dtype=torch.int64, device=self._flat_counters.device)
            used = []
            for key, shape, kind in self._spec:
                if kind == "counter":
                    top = key.split(".")[0]
                    u = True
                    if top in ("convOne", "convAvgRest") and not self.interaction:
                        u = False
                    if top in getattr(self, "_frozen_children", ()):     # eval() mode: num_batches_tracked stands still
                        u = False
                    used.append(1 if u else 0)
            self._counter_inc = self._counter_inc * torch.tensor(used, dtype=torch.int64,
                                                                 device=self._flat_counters.device)
        self._flat_counters += self._counter_inc

    def forward(self, x, return_featuremaps=False):
        """x: list/tuple of three [B,3,H,W] tensors in the order [RGB, NI, TI].  The second positional
        argument is accepted and ignored: the reference's engine passes `timeids` there
        (engine/engine.py:366, 450-451)."""
        assert isinstance(x, (list, tuple)) and len(x) == 3, "expected [RGB, NI, TI]"
        B, C, H, W = x[0].shape
        assert C == 3
        net = self.native_net(B, H, W)
        if not self.training:
            with torch.no_grad():
                _, fc_all = net.forward(x, training=False)
            return fc_all
        self._bump_counters()
        if torch.is_grad_enabled():
            params = [p for _, p in self._param_items]
            logits, feats = _NetFunction.apply(self, net, x[0], x[1], x[2], *params)
        else:
            logits, feats = net.forward(x, training=True)
        result_R = [logits[i] for i in range(6)]
        result_N = [logits[6 + i] for i in range(6)]
        result_T = [logits[12 + i] for i in range(6)]
        if self.loss == 'softmax':
            return result_R, result_N, result_T
        elif self.loss == 'margin':
            return result_R, result_N, result_T, feats[0], feats[1], feats[2]
        raise KeyError("loss '{}' is not live for IEEE3modalPart in the reference either "
                       "(only 'softmax' and 'margin' engines accept its outputs)".format(self.loss))


def ieee3modalPart(num_classes, loss='softmax', pretrained=True, **kwargs):
    """factory, reference ieee3modalPart.py:542-555"""
    return IEEE3modalPart(num_classes=num_classes, loss=loss, block=None, layers=[3, 4, 6, 3], last_stride=1,
                          parts=6, reduced_dim=768, nonlinear='relu', pretrained=pretrained, **kwargs)


__model_factory = {'ieee3modalPart': ieee3modalPart}


def show_avai_models():
    print(list(__model_factory.keys()))


def build_model(name, num_classes, loss='softmax', pretrained=True, use_gpu=True, **kwargs):
    """reference torchreid/models/__init__.py:80-111 (only the hot-path model is registered)"""
    avai_models = list(__model_factory.keys())
    if name not in avai_models:
        raise KeyError('Unknown model: {}. Must be one of {}'.format(name, avai_models))
    return __model_factory[name](num_classes=num_classes, loss=loss, pretrained=pretrained, use_gpu=use_gpu, **kwargs)
