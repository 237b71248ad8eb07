"""Batches for the engines: CPU workers only decode JPEGs (the reference runs its whole transform chain in the loader
process with workers=0, configs/RGBNT_ieee_part_margin.yaml:13); the decoded bytes are resized / flipped / normalised on
the GPU per batch.  Yields the reference's batch dict: {'img': [RGB, NI, TI] float tensors [B,3,H,W] (on the device),
'pid', 'camid', 'impath', 'timeid'} (data/datasets/dataset.py:344-351)."""
import torch
from torch.utils.data import DataLoader

from .datasets import MultiModalImageDataset
from .sampler import build_train_sampler
from .transforms import build_transforms


def _collate(items):
    return {'img': [it['img'] for it in items],                       # [sample][modality] uint8 arrays
            'pid': torch.as_tensor([it['pid'] for it in items], dtype=torch.int64),
            'camid': torch.as_tensor([it['camid'] for it in items], dtype=torch.int64),
            'impath': [it['impath'] for it in items],
            'timeid': torch.as_tensor([it['timeid'] for it in items], dtype=torch.int64)}


class DeviceLoader(object):
    def __init__(self, data, transform, batch_size, sampler=None, shuffle=False, workers=4, drop_last=False):
        self.dataset = MultiModalImageDataset(data)
        self.transform = transform
        self.loader = DataLoader(self.dataset, batch_size=batch_size, sampler=sampler, shuffle=shuffle and sampler is None,
                                 num_workers=workers, collate_fn=_collate, drop_last=drop_last, pin_memory=False)

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for batch in self.loader:
            raw = batch['img']
            n, mods = len(raw), len(raw[0])
            # the reference transforms sample by sample, modality by modality: draw the flips in that order
            flips = self.transform.draw_flips(n * mods).reshape(n, mods)
            batch['img'] = [self.transform([raw[i][m] for i in range(n)], flips=flips[:, m]) for m in range(mods)]
            yield batch


def build_loaders(dataset, height=256, width=128, transforms='random_flip', batch_size_train=8, batch_size_test=100,
                  train_sampler='RandomIdentitySampler', num_instances=4, workers=4, norm_mean=None, norm_std=None):
    """train / query / gallery loaders for an RGBNT201-style dataset object (reference data/datamanager.py:158-245)"""
    tr, te = build_transforms(height, width, transforms, norm_mean, norm_std)
    sampler = build_train_sampler(dataset.train, train_sampler, batch_size=batch_size_train, num_instances=num_instances)
    train = DeviceLoader(dataset.train, tr, batch_size_train, sampler=sampler, workers=workers, drop_last=True)
    query = DeviceLoader(dataset.query, te, batch_size_test, workers=workers)
    gallery = DeviceLoader(dataset.gallery, te, batch_size_test, workers=workers)
    return train, query, gallery
