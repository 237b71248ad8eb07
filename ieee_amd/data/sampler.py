"""Identity-balanced batch order (the behaviour of the reference's RandomIdentitySampler, torchreid/data/sampler.py:17-84).

Contract kept: every batch holds P = batch_size // K identities with K = num_instances consecutive samples each --
the contiguity the 3M loss's chunk() relies on and that ieee_amd.dist.shard_bounds uses to cut a global batch on
identity boundaries -- and, for the same `random` / `numpy.random` seeds, the index sequence is the reference's
(tests/test_data_cpu.py compares against the imported reference class).  That pins the ORDER of the random draws, not
the code: per identity (first-seen order) one optional `numpy.random.choice` top-up and one `random.shuffle`, then one
`random.sample` of P identities per batch from the identities that still have a group left.
"""
import random
from collections import OrderedDict, deque

import numpy as np
from torch.utils.data.sampler import RandomSampler, Sampler, SequentialSampler

from ..dist import shard_bounds

AVAI_SAMPLERS = ['RandomIdentitySampler', 'SequentialSampler', 'RandomSampler']


class RandomIdentitySampler(Sampler):
    """data_source: sequence of (paths, pid, camid, ...) records; yields dataset indices"""

    def __init__(self, data_source, batch_size, num_instances):
        if batch_size < num_instances:
            raise ValueError('batch_size={} must be no less than num_instances={}'.format(batch_size, num_instances))
        self.data_source = data_source
        self.batch_size = int(batch_size)
        self.num_instances = K = int(num_instances)
        self.ids_per_batch = self.batch_size // K
        members = OrderedDict()                                   # pid -> dataset indices, identities in first-seen order
        for index, record in enumerate(data_source):
            members.setdefault(record[1], []).append(index)
        self._members = members
        if len(members) < self.ids_per_batch:
            raise AssertionError('{} identities cannot fill a batch of {}'.format(len(members), self.ids_per_batch))
        # samples per epoch if every identity's groups were all used: whole groups of K, at least one per identity
        self._epoch_len = sum(max(len(v), K) // K * K for v in members.values())

    def _groups_of(self, indices):
        """one identity's shuffled indices cut into whole groups of K (a short identity is topped up with repeats first)"""
        K = self.num_instances
        pool = list(indices)
        if len(pool) < K:
            pool = np.random.choice(pool, size=K, replace=True)   # draw 1 (only for short identities)
        random.shuffle(pool)                                      # draw 2
        pool = [int(i) for i in pool]
        return deque(pool[g:g + K] for g in range(0, len(pool) - len(pool) % K, K))

    def __iter__(self):
        queues = OrderedDict((pid, self._groups_of(idx)) for pid, idx in self._members.items())
        alive = list(queues)                                      # identities with a group left, first-seen order
        order = []
        while len(alive) >= self.ids_per_batch:
            for pid in random.sample(alive, self.ids_per_batch):  # draw 3: this batch's identities, in this order
                order += queues[pid].popleft()
                if not queues[pid]:
                    alive.remove(pid)
        return iter(order)

    def __len__(self):
        return self._epoch_len


class ShardedIdentitySampler(Sampler):
    """Rank `rank`'s identity-aligned slice of every GLOBAL batch of a RandomIdentitySampler (the
    reference's nn.DataParallel scatters each batch over the GPUs, scripts/mainMultiModal.py:219-220; one process per
    GPU draws only its own rows instead, so a rank decodes and transforms B/world triples per step, not B).

    The global index sequence is the single-process one: rank 0 draws it exactly as RandomIdentitySampler does (same
    `random` / `numpy.random` seeds => the reference's batches) and, when torch.distributed is initialised, broadcasts
    it, so the ranks agree even if their seeds do not; without a process group (tests, `world` given explicitly) every
    rank draws it itself and the caller seeds them alike.  Global batch g = order[g*B:(g+1)*B]; this rank keeps rows
    [a, b) = dist.shard_bounds(B, K, world, rank) of it -- whole identities, so every 3M chunk stays rank-local.  Feed it
    to a DataLoader with batch_size = local_batch and drop_last."""

    def __init__(self, base, rank, world):
        if not isinstance(base, RandomIdentitySampler):
            raise TypeError('ShardedIdentitySampler shards a RandomIdentitySampler')
        if not 0 <= rank < world:
            raise ValueError('rank {} outside world {}'.format(rank, world))
        self.base, self.rank, self.world = base, int(rank), int(world)
        self.global_batch = base.ids_per_batch * base.num_instances     # what a batch of the base sampler really holds
        self.lo, self.hi = shard_bounds(self.global_batch, base.num_instances, self.world, self.rank)
        self.local_batch = self.hi - self.lo
        if self.local_batch == 0:
            raise ValueError('{} identities per batch cannot feed {} ranks'.format(base.ids_per_batch, world))

        self._prepared = None      # the order drawn by prepare(), consumed by the next __iter__
        self._last_len = None      # samples the last drawn epoch really yielded on this rank

    def _live_group(self):
        import torch.distributed as dist
        live = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        if live and (dist.get_world_size() != self.world or dist.get_rank() != self.rank):
            # the shard bounds were computed for (self.rank, self.world): a group of another shape would make rank 0 of
            # the group draw for a sampler that slices by a different rank / world -- refuse instead of disagreeing silently
            raise RuntimeError('ShardedIdentitySampler(rank={}, world={}) inside a process group of rank {} / world {}'.format(
                self.rank, self.world, dist.get_rank(), dist.get_world_size()))
        return live

    def global_order(self):
        """the epoch's global index sequence: drawn by rank 0 and broadcast when a process group is live (a COLLECTIVE:
        every rank must call it, from its main process, with its device already selected under NCCL), else drawn locally"""
        import torch.distributed as dist
        if not self._live_group():
            return list(iter(self.base))
        box = [list(iter(self.base)) if self.rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    def prepare(self):
        """Draw (and, with a live process group, exchange) the next epoch's order NOW, in the caller's context, instead of
        inside __iter__ -- where a DataLoader would trigger the collective at its first prefetch.  DeviceLoader calls this
        at the start of every epoch; set_epoch() is the torch DistributedSampler spelling of the same call."""
        self._prepared = self._slice(self.global_order())
        return self

    def set_epoch(self, epoch=None):
        return self.prepare()

    def _slice(self, order):
        B = self.global_batch
        mine = []
        for start in range(0, len(order) - len(order) % B, B):
            mine += order[start + self.lo:start + self.hi]
        self._last_len = len(mine)
        return mine

    def __iter__(self):
        mine, self._prepared = self._prepared, None
        if mine is None:
            mine = self._slice(self.global_order())
        return iter(mine)

    def __len__(self):
        """samples of the epoch drawn last (exact); before any draw, the base sampler's figure -- an UPPER bound: the
        identity sampler stops when fewer than P identities have a group left (the reference's __len__ over-reports the
        same way, sampler.py:81-84)"""
        if self._last_len is not None:
            return self._last_len
        return len(self.base) // self.global_batch * self.local_batch


def build_train_sampler(data_source, train_sampler, batch_size=32, num_instances=4, rank=0, world=1, **kwargs):
    """the samplers the 3-modal configs can name (reference sampler.py:216-255); world > 1: this rank's shard of the
    identity sampler's batches (batch_size is the GLOBAL batch)"""
    if train_sampler not in AVAI_SAMPLERS:
        raise AssertionError('train_sampler must be one of {}, but got {}'.format(AVAI_SAMPLERS, train_sampler))
    if train_sampler == 'RandomIdentitySampler':
        base = RandomIdentitySampler(data_source, batch_size, num_instances)
        return base if world == 1 else ShardedIdentitySampler(base, rank, world)
    if world > 1:
        raise ValueError('data-parallel training shards on identity boundaries: use RandomIdentitySampler')
    return SequentialSampler(data_source) if train_sampler == 'SequentialSampler' else RandomSampler(data_source)
