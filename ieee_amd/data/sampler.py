"""Identity sampler, reference torchreid/data/sampler.py:17-84 (same use of the `random` / `numpy.random` streams, so
the same seeds give the same index sequence): every batch holds batch_size/num_instances identities with
num_instances consecutive samples each -- the contiguity the 3M loss's chunk() relies on and that
ieee_amd.dist.shard_bounds uses to cut a global batch on identity boundaries."""
import copy
import random
from collections import defaultdict

import numpy as np
from torch.utils.data.sampler import RandomSampler, Sampler, SequentialSampler

AVAI_SAMPLERS = ['RandomIdentitySampler', 'SequentialSampler', 'RandomSampler']


class RandomIdentitySampler(Sampler):
    def __init__(self, data_source, batch_size, num_instances):
        if batch_size < num_instances:
            raise ValueError('batch_size={} must be no less than num_instances={}'.format(batch_size, num_instances))
        self.data_source = data_source
        self.batch_size = batch_size
        self.num_instances = num_instances
        self.num_pids_per_batch = self.batch_size // self.num_instances
        self.index_dic = defaultdict(list)
        for index, items in enumerate(data_source):
            self.index_dic[items[1]].append(index)
        self.pids = list(self.index_dic.keys())
        assert len(self.pids) >= self.num_pids_per_batch
        self.length = 0                              # estimate of the examples per epoch (sampler.py:42-50)
        for pid in self.pids:
            num = max(len(self.index_dic[pid]), self.num_instances)
            self.length += num - num % self.num_instances

    def __iter__(self):
        batch_idxs_dict = defaultdict(list)
        for pid in self.pids:
            idxs = copy.deepcopy(self.index_dic[pid])
            if len(idxs) < self.num_instances:
                idxs = np.random.choice(idxs, size=self.num_instances, replace=True)
            random.shuffle(idxs)
            batch_idxs = []
            for idx in idxs:
                batch_idxs.append(idx)
                if len(batch_idxs) == self.num_instances:
                    batch_idxs_dict[pid].append(batch_idxs)
                    batch_idxs = []
        avai_pids = copy.deepcopy(self.pids)
        final_idxs = []
        while len(avai_pids) >= self.num_pids_per_batch:
            selected_pids = random.sample(avai_pids, self.num_pids_per_batch)
            for pid in selected_pids:
                batch_idxs = batch_idxs_dict[pid].pop(0)
                final_idxs.extend(batch_idxs)
                if len(batch_idxs_dict[pid]) == 0:
                    avai_pids.remove(pid)
        return iter(final_idxs)

    def __len__(self):
        return self.length


def build_train_sampler(data_source, train_sampler, batch_size=32, num_instances=4, **kwargs):
    """reference sampler.py:216-255 (the samplers the 3-modal configs use)"""
    assert train_sampler in AVAI_SAMPLERS, 'train_sampler must be one of {}, but got {}'.format(AVAI_SAMPLERS, train_sampler)
    if train_sampler == 'RandomIdentitySampler':
        return RandomIdentitySampler(data_source, batch_size, num_instances)
    if train_sampler == 'SequentialSampler':
        return SequentialSampler(data_source)
    return RandomSampler(data_source)
