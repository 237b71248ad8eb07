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
    """rank / world (evaluation loaders): batch b belongs to rank b % world; only the owned batches
    are decoded and transformed -- the other ranks' rows never leave the disk here -- and every yielded batch says which
    global batch it is (`batch_index`).  `batch_labels()` lists (pids, camids) of ALL batches from the dataset records, so
    the evaluator can lay out the gathered descriptor matrix without touching an image.
    global_rows (training loader fed by a ShardedIdentitySampler): the size of the global batch this rank's batches are
    shards of, stamped on every batch so the engine needs no collective to scale the cross entropy."""

    def __init__(self, data, transform, batch_size, sampler=None, shuffle=False, workers=4, drop_last=False, rank=0, world=1,
                 global_rows=None):
        self.dataset = MultiModalImageDataset(data)
        self.transform = transform
        self.rank, self.world = int(rank), int(world)
        self.global_rows = global_rows
        self.batch_size = int(batch_size)
        self._owned = None
        if self.world > 1 and sampler is None and not shuffle:
            n = len(self.dataset)
            self._all = [list(range(s, min(s + self.batch_size, n))) for s in range(0, n, self.batch_size)]
            if drop_last and self._all and len(self._all[-1]) < self.batch_size:
                self._all.pop()
            self._owned = list(range(self.rank, len(self._all), self.world))
            self.loader = DataLoader(self.dataset, batch_sampler=[self._all[b] for b in self._owned], num_workers=workers,
                                     collate_fn=_collate, pin_memory=False)
        else:
            self.loader = DataLoader(self.dataset, batch_size=batch_size, sampler=sampler, shuffle=shuffle and sampler is None,
                                     num_workers=workers, collate_fn=_collate, drop_last=drop_last, pin_memory=False)

    @property
    def sharded(self):
        return self._owned is not None

    def batch_labels(self):
        """[(pids, camids)] of every global batch, in loader order, from the records alone (no decode)"""
        assert self._owned is not None, "batch_labels() is for rank-sharded sequential loaders"
        recs = self.dataset.data
        return [([recs[i][1] for i in idx], [recs[i][2] for i in idx]) for idx in self._all]

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        sampler = getattr(self.loader, "sampler", None)
        if hasattr(sampler, "prepare"):          # rank-sharded identity sampler: draw / exchange the epoch's order here, in
            sampler.prepare()                    # the main process, not at the DataLoader's first prefetch
        lo, hi = getattr(sampler, "lo", None), getattr(sampler, "hi", None)
        for k, batch in enumerate(self.loader):
            raw = batch['img']
            n, mods = len(raw), len(raw[0])
            # the reference transforms sample by sample, modality by modality: draw the flips in that order
            if self.global_rows is not None and lo is not None and hi - lo == n:
                # a shard of a global batch: draw the flips of the WHOLE global batch -- every rank consumes the same
                # torch RNG stream, as the single-process loop would -- and keep this shard's rows; ranks seeded alike
                # (torch.manual_seed, as the reference's set_random_seed does) then augment the global batch exactly
                # like the single-process sequence instead of every shard repeating rank 0's pattern
                flips = self.transform.draw_flips(int(self.global_rows) * mods).reshape(int(self.global_rows), mods)[lo:hi]
            else:
                flips = self.transform.draw_flips(n * mods).reshape(n, mods)
            batch['img'] = [self.transform([raw[i][m] for i in range(n)], flips=flips[:, m]) for m in range(mods)]
            if self._owned is not None:
                batch['batch_index'] = self._owned[k]
            if self.global_rows is not None:
                batch['global_rows'] = int(self.global_rows)
            yield batch


def build_loaders(dataset, height=256, width=128, transforms='random_flip', batch_size_train=8, batch_size_test=100,
                  train_sampler='RandomIdentitySampler', num_instances=4, workers=4, norm_mean=None, norm_std=None,
                  rank=None, world=None):
    """train / query / gallery loaders for an RGBNT201-style dataset object (reference data/datamanager.py:158-245).
    rank / world default to the process group's (ieee_amd.dist).  With several ranks batch_size_train is the GLOBAL batch:
    the train loader yields this rank's identity-aligned shard of every global batch (ShardedIdentitySampler; the engine
    sees `global_rows` in the batch and skips its own slicing), and the query / gallery loaders decode every world-th batch."""
    from .. import dist as ddp
    world = ddp.world_size() if world is None else int(world)
    rank = ddp.rank() if rank is None else int(rank)
    tr, te = build_transforms(height, width, transforms, norm_mean, norm_std)
    sampler = build_train_sampler(dataset.train, train_sampler, batch_size=batch_size_train, num_instances=num_instances,
                                  rank=rank, world=world)
    if world > 1:
        train = DeviceLoader(dataset.train, tr, sampler.local_batch, sampler=sampler, workers=workers, drop_last=True,
                             global_rows=sampler.global_batch)
    else:
        train = DeviceLoader(dataset.train, tr, batch_size_train, sampler=sampler, workers=workers, drop_last=True)
    query = DeviceLoader(dataset.query, te, batch_size_test, workers=workers, rank=rank, world=world)
    gallery = DeviceLoader(dataset.gallery, te, batch_size_test, workers=workers, rank=rank, world=world)
    return train, query, gallery
