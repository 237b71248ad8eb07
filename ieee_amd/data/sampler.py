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


def build_train_sampler(data_source, train_sampler, batch_size=32, num_instances=4, **kwargs):
    """the samplers the 3-modal configs can name (reference sampler.py:216-255)"""
    if train_sampler not in AVAI_SAMPLERS:
        raise AssertionError('train_sampler must be one of {}, but got {}'.format(AVAI_SAMPLERS, train_sampler))
    if train_sampler == 'RandomIdentitySampler':
        return RandomIdentitySampler(data_source, batch_size, num_instances)
    return SequentialSampler(data_source) if train_sampler == 'SequentialSampler' else RandomSampler(data_source)
