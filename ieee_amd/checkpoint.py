"""Checkpoint / pretrained-weight interop for the flat-buffer model.

Mirrors the reference's torchreid/utils/torchtools.py (save_checkpoint :20-58, load_checkpoint :61-95,
resume_from_checkpoint :98-133, load_pretrained_weights :256-311) and the backbone initialisation of
torchreid/models/resnet.py:1075-1089 (init_pretrained_weights: an ImageNet ResNet-50 state_dict is copied into a
backbone wherever name AND shape match, everything else is left as it is).  Because IEEE3modalPart's state_dict here
has the reference's 1197 keys in the reference's order (tests/golden/state_dict_spec.txt), a `model.pth.tar-N` written
by either implementation loads in the other, with or without DataParallel's "module." prefix."""
import os
import os.path as osp
import pickle
import warnings
from collections import OrderedDict
from functools import partial

import torch


RESNET50_FILE = "resnet50-19c8e357.pth"      # the file the reference downloads (resnet.py:25-26)


def save_checkpoint(state, save_dir, is_best=False, remove_module_from_keys=False):
    """Writes `state` (keys as in the reference: state_dict, epoch, mAP / rank1, optimizer, scheduler) to
    <save_dir>/model.pth.tar-<epoch>, the file name resume_from_checkpoint / load_pretrained_weights of either
    implementation expect (reference torchtools.py:20-58); is_best also leaves a copy as model-best.pth.tar."""
    os.makedirs(save_dir, exist_ok=True)
    if remove_module_from_keys:        # weights saved from inside nn.DataParallel
        state = dict(state)
        state['state_dict'] = OrderedDict((name[len('module.'):] if name.startswith('module.') else name, tensor)
                                          for name, tensor in state['state_dict'].items())
    target = osp.join(save_dir, 'model.pth.tar-%s' % state['epoch'])
    torch.save(state, target)
    print('Checkpoint saved to "{}"'.format(target))
    if is_best:
        import shutil
        shutil.copy(target, osp.join(save_dir, 'model-best.pth.tar'))
    return target


def _torch_load(path, **extra):
    device = None if torch.cuda.is_available() else 'cpu'      # CPU-only hosts can open GPU-written files
    return torch.load(path, map_location=device, weights_only=False, **extra)


def load_checkpoint(fpath):
    """Reads a checkpoint file written by either implementation (reference torchtools.py:61-95): ValueError for a
    missing argument, FileNotFoundError for a missing file; a python-2 pickle is retried with latin-1 decoding."""
    if fpath is None:
        raise ValueError('no checkpoint path given (fpath is None)')
    if not osp.isfile(fpath):
        raise FileNotFoundError('no checkpoint file at "{}"'.format(fpath))
    try:
        return _torch_load(fpath)
    except UnicodeDecodeError:
        legacy = type(pickle)('pickle_latin1')                 # a pickle look-alike module that decodes py2 strings
        legacy.__dict__.update(pickle.__dict__)
        legacy.load = partial(pickle.load, encoding='latin1')
        legacy.Unpickler = partial(pickle.Unpickler, encoding='latin1')
        return _torch_load(fpath, pickle_module=legacy)
    except Exception:
        print('could not read the checkpoint "{}"'.format(fpath))
        raise


def resume_from_checkpoint(fpath, model, optimizer=None, scheduler=None):
    """Restores model (and, when given and present, optimizer / scheduler) state and returns the epoch to continue
    from (reference torchtools.py:98-133; same checkpoint keys: state_dict, optimizer, scheduler, epoch, rank1)."""
    ckpt = load_checkpoint(fpath)
    restored = ['model']
    model.load_state_dict(ckpt['state_dict'])
    for name, target in (('optimizer', optimizer), ('scheduler', scheduler)):
        if target is not None and name in ckpt:
            target.load_state_dict(ckpt[name])
            restored.append(name)
    epoch = ckpt['epoch']
    note = ', rank-1 then {:.1%}'.format(ckpt['rank1']) if 'rank1' in ckpt else ''
    print('resumed {} from "{}": epoch {}{}'.format(' + '.join(restored), fpath, epoch, note))
    return epoch


def _matching(model_dict, state_dict, prefix=''):
    matched, discarded = OrderedDict(), []
    for k, v in state_dict.items():
        if k.startswith('module.'):
            k = k[7:]                                  # DataParallel prefix
        key = prefix + k
        if key in model_dict and tuple(model_dict[key].shape) == tuple(v.shape):
            matched[key] = v
        else:
            discarded.append(k)
    return matched, discarded


def load_pretrained_weights(model, weight_path):
    """torchtools.py:256-311: layers unmatched in name or size are ignored, "module." is stripped"""
    checkpoint = load_checkpoint(weight_path)
    state_dict = checkpoint['state_dict'] if 'state_dict' in checkpoint else checkpoint
    model_dict = model.state_dict()
    matched, discarded = _matching(model_dict, state_dict)
    model_dict.update(matched)
    model.load_state_dict(model_dict)
    if len(matched) == 0:
        warnings.warn('The pretrained weights "{}" cannot be loaded, please check the key names manually '
                      '(** ignored and continue **)'.format(weight_path))
    else:
        print('Successfully loaded pretrained weights from "{}"'.format(weight_path))
        if len(discarded) > 0:
            print('** The following layers are discarded due to unmatched keys or layer size: {}'.format(discarded))
    return list(matched), discarded


def find_resnet50_file():
    """where torch.hub / model_zoo would have cached the reference's download, or $IEEE_RESNET50_PTH"""
    cands = [os.environ.get("IEEE_RESNET50_PTH")]
    hub = os.environ.get("TORCH_HOME", osp.join(osp.expanduser("~"), ".cache", "torch"))
    cands += [osp.join(hub, "hub", "checkpoints", RESNET50_FILE), osp.join(hub, "checkpoints", RESNET50_FILE)]
    for c in cands:
        if c and osp.exists(c):
            return c
    return None


def init_pretrained_backbones(model, resnet50_state):
    """resnet.py:1075-1089 applied to the three backbones (ieee3modalPart.py:305-310 builds each with
    pretrained=True): every ImageNet tensor whose name and shape match `backbone.{m}.<name>` is copied into all three
    modality streams; `fc.*` (1000 classes) has no counterpart and is dropped, as in the reference.
    resnet50_state: a state_dict or a path to resnet50-19c8e357.pth.  Returns the number of tensors set."""
    if isinstance(resnet50_state, str):
        resnet50_state = load_checkpoint(resnet50_state)
    model_dict = model.state_dict()
    n = 0
    for m in range(3):
        matched, _ = _matching(model_dict, resnet50_state, prefix='backbone.%d.' % m)
        model_dict.update(matched)
        n += len(matched)
    model.load_state_dict(model_dict)
    return n
