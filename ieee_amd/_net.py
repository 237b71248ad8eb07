"""Host-side handle of the native layer-graph executor (ieee_net_* in include/ieee_amd.h)."""
import ctypes

import torch

from . import _lib


class NativeNet:
    """One executor instance = fixed (batch, image size, dtype, ablation flags).  Owns the workspace
    tensor; parameters / gradients / running statistics stay in the model's flat buffers."""

    def __init__(self, model, batch, height, width, dtype):
        lib = _lib.require_gpu()
        self.lib = lib
        self.batch, self.height, self.width, self.dtype = batch, height, width, dtype
        self.num_classes = model.num_classes
        self.handle = ctypes.c_void_p()
        dt = _lib.IEEE_BF16 if dtype == torch.bfloat16 else _lib.IEEE_F32
        _lib.check(lib.ieee_net_create(batch, height, width, model.num_classes, dt, int(model.interaction),
                                       int(model.attention), int(model.using_REM), ctypes.byref(self.handle)))
        n = lib.ieee_net_num_slots(self.handle)
        names = [lib.ieee_net_slot_name(self.handle, i).decode() for i in range(n)]
        offs = (ctypes.c_int64 * n)()
        for i, name in enumerate(names):
            if name not in model._offsets:
                raise KeyError("executor needs tensor %r which the model does not own" % name)
            offs[i] = model._offsets[name]
        self._flat = (model._flat_params, model._flat_grads, model._flat_buffers)   # keep alive
        _lib.check(lib.ieee_net_bind(self.handle, _lib.ptr(model._flat_params), _lib.ptr(model._flat_grads),
                                     _lib.ptr(model._flat_buffers), offs, n))
        nbytes = lib.ieee_net_workspace_bytes(self.handle)
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=model._flat_params.device)
        self.model = model
        self._eval_key = None
        self._shadow_set = False

    def __del__(self):
        try:
            if self.handle:
                self.lib.ieee_net_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def forward(self, xs, training):
        B = self.batch
        dev = self.workspace.device
        xs = [x.to(device=dev, dtype=torch.float32).contiguous() for x in xs]
        if training:
            logits = torch.empty((18, B, self.num_classes), dtype=torch.float32, device=dev)
            feats = torch.empty((3, B, 768), dtype=torch.float32, device=dev)
        else:
            logits = None
            feats = torch.empty((B, 2304), dtype=torch.float32, device=dev)
        m = self.model
        # inference cache (ieee_net_eval_cache): valid while nothing has written the parameters / running statistics --
        # torch's version counters see every in-place op on the views, `_native_epoch` counts the native writers
        key = (m._flat_params._version, m._flat_buffers._version, m._native_epoch)
        if training:
            # the 1x1 convolutions' forward operands straight from the optimizer's bf16 shadow of the parameters (when one is kept)
            use = getattr(m, "_shadow_enabled", False) and self.dtype == torch.bfloat16
            sh = m.fresh_shadow() if use else None
            if use or self._shadow_set:
                _lib.check(self.lib.ieee_net_set_shadow(self.handle, _lib.ptr(sh)))
                self._shadow_set = use
            m._native_epoch += 1                      # running statistics are updated by the native forward
            if use and m._shadow_key is not None:     # (that bump is not a parameter write: the shadow stays current)
                m._shadow_key = (m._shadow_key[0], m._native_epoch)
        elif key != self._eval_key:
            _lib.check(self.lib.ieee_net_eval_cache(self.handle, 0))
        _lib.check(self.lib.ieee_net_forward(self.handle, _lib.ptr(self.workspace), _lib.ptr(xs[0]), _lib.ptr(xs[1]),
                                             _lib.ptr(xs[2]), 1 if training else 0, _lib.ptr(logits),
                                             _lib.ptr(feats), _lib.stream()))
        self._eval_key = None if training else key
        return logits, feats

    def backward(self, dlogits, dfeats):
        dlogits = dlogits.to(torch.float32).contiguous()
        dfeats = dfeats.to(torch.float32).contiguous()
        _lib.check(self.lib.ieee_net_backward(self.handle, _lib.ptr(self.workspace), _lib.ptr(dlogits),
                                              _lib.ptr(dfeats), _lib.stream()))

    def backward_part(self, dlogits, dfeats, part):
        """staged backward (include/ieee_amd.h: ieee_net_backward_part); parts 0..4 in order"""
        _lib.check(self.lib.ieee_net_backward_part(self.handle, _lib.ptr(self.workspace), _lib.ptr(dlogits),
                                                   _lib.ptr(dfeats), part, _lib.stream()))

    def backward_part_async(self, dlogits, dfeats, part):
        """like backward_part, but the launch stream is not made to wait for the part's weight gradients (they run on
        the executor's side stream); pair with side_wait()"""
        _lib.check(self.lib.ieee_net_backward_part_async(self.handle, _lib.ptr(self.workspace), _lib.ptr(dlogits),
                                                         _lib.ptr(dfeats), part, _lib.stream()))

    def side_wait(self, stream=None):
        """stream=None: final join on the current (launch) stream; else make that torch stream wait for every weight
        gradient issued so far without blocking the launch stream"""
        if stream is None:
            _lib.check(self.lib.ieee_net_side_wait(self.handle, _lib.ptr(self.workspace), _lib.stream(), 1))
        else:
            _lib.check(self.lib.ieee_net_side_wait(self.handle, _lib.ptr(self.workspace),
                                                   ctypes.c_void_p(stream.cuda_stream), 0))

    def set_frozen(self, mask):
        """children whose BatchNorms run on their running statistics in a training forward (ieee_net_set_frozen)"""
        _lib.check(self.lib.ieee_net_set_frozen(self.handle, int(mask)))

    def bn_overflow(self):
        """(fwd tile clamped, bwd tile clamped, fwd total beyond half the range, bwd total beyond half the range) reported by
        the range guard of the fixed-point BatchNorm totals during the most recent training step (ieee_net_bn_overflow: a
        blocking read of the device words, which it clears; the caller has synchronised with the step it asks about)"""
        out = (ctypes.c_int * 4)()
        _lib.check(self.lib.ieee_net_bn_overflow(self.handle, out))
        return tuple(out)

    def flags_view(self):
        """the four int32 range-guard words of this executor as a torch view of the workspace (device memory, zeroed by every
        training forward; [0] / [1]: a forward / backward BatchNorm tile sum of the step was clamped)"""
        if getattr(self, "_flags", None) is None:
            off = int(self.lib.ieee_net_bn_flags_offset(self.handle))
            self._flags = self.workspace[off:off + 16].view(torch.int32)
        return self._flags

    def set_bn_totals(self, on):
        """on=False: every BatchNorm on the per-tile partial-sum path from the next forward on (ieee_net_set_bn_totals)"""
        _lib.check(self.lib.ieee_net_set_bn_totals(self.handle, 1 if on else 0))

    def debug_taps(self, nbytes):
        """parity tests: allocate a tap buffer of nbytes and make the backward copy its gradient tensors into it
        (include/ieee_amd.h: ieee_net_debug_taps); nbytes = 0 switches the taps off"""
        self._taps = torch.empty(nbytes, dtype=torch.uint8, device=self.workspace.device) if nbytes else None
        _lib.check(self.lib.ieee_net_debug_taps(self.handle, _lib.ptr(self._taps) if nbytes else None, nbytes))

    def tap(self, name):
        off, numel, dt = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
        _lib.check(self.lib.ieee_net_debug_tap(self.handle, name.encode(), ctypes.byref(off), ctypes.byref(numel),
                                               ctypes.byref(dt)))
        tdt = {0: torch.float32, 1: torch.bfloat16, 2: torch.uint8}[dt.value]
        nbytes = numel.value * torch.empty((), dtype=tdt).element_size()
        return self._taps[off.value:off.value + nbytes].view(tdt)

    def tensor(self, name):
        """a named intermediate as a torch view of the workspace (parity tests / debugging)"""
        off, numel, dt = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
        _lib.check(self.lib.ieee_net_tensor(self.handle, name.encode(), ctypes.byref(off), ctypes.byref(numel),
                                            ctypes.byref(dt)))
        tdt = {0: torch.float32, 1: torch.bfloat16, 2: torch.uint8}[dt.value]
        nbytes = numel.value * torch.empty((), dtype=tdt).element_size()
        return self.workspace[off.value:off.value + nbytes].view(tdt)
