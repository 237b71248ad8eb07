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

from .engine import save_checkpoint   # noqa: F401  (same function, reference torchtools.py:20-58)

RESNET50_FILE = "resnet50-19c8e357.pth"      # the file the reference downloads (resnet.py:25-26)


def load_checkpoint(fpath):
    """torchtools.py:61-95 (python2 pickles handled the same way)"""
    if fpath is None:
        raise ValueError('File path is None')
    if not osp.exists(fpath):
        raise FileNotFoundError('File is not found at "{}"'.format(fpath))
    map_location = None if torch.cuda.is_available() else 'cpu'
    try:
        checkpoint = torch.load(fpath, map_location=map_location, weights_only=False)
    except UnicodeDecodeError:
        pickle.load = partial(pickle.load, encoding="latin1")
        pickle.Unpickler = partial(pickle.Unpickler, encoding="latin1")
        checkpoint = torch.load(fpath, pickle_module=pickle, map_location=map_location, weights_only=False)
    except Exception:
        print('Unable to load checkpoint from "{}"'.format(fpath))
        raise
    return checkpoint


def resume_from_checkpoint(fpath, model, optimizer=None, scheduler=None):
    """torchtools.py:98-133; returns start_epoch"""
    print('Loading checkpoint from "{}"'.format(fpath))
    checkpoint = load_checkpoint(fpath)
    model.load_state_dict(checkpoint['state_dict'])
    print('Loaded model weights')
    if optimizer is not None and 'optimizer' in checkpoint.keys():
        optimizer.load_state_dict(checkpoint['optimizer'])
        print('Loaded optimizer')
    if scheduler is not None and 'scheduler' in checkpoint.keys():
        scheduler.load_state_dict(checkpoint['scheduler'])
        print('Loaded scheduler')
    start_epoch = checkpoint['epoch']
    print('Last epoch = {}'.format(start_epoch))
    if 'rank1' in checkpoint.keys():
        print('Last rank1 = {:.1%}'.format(checkpoint['rank1']))
    return start_epoch


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
