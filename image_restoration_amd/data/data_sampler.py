"""Rank-strided, epoch-seeded sampler with dataset enlargement.

Same index streams as basicsr/data/data_sampler.py:6-48: ``randperm(total_size, seed=epoch) % len(dataset)`` then
``[rank::num_replicas]`` — data-parallel ranks see disjoint, reproducible shards."""
import math

import torch
from torch.utils.data.sampler import Sampler


class EnlargedSampler(Sampler):
    """EnlargedSampler(dataset, num_replicas, rank, ratio=1)."""

    def __init__(self, dataset, num_replicas, rank, ratio=1):
        self.dataset, self.num_replicas, self.rank, self.epoch = dataset, num_replicas, rank, 0
        self.num_samples = math.ceil(len(dataset) * ratio / num_replicas)
        self.total_size = self.num_samples * num_replicas

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return self.num_samples

    def __iter__(self):
        gen = torch.Generator().manual_seed(self.epoch)
        n = len(self.dataset)
        order = [i % n for i in torch.randperm(self.total_size, generator=gen).tolist()]
        mine = order[self.rank::self.num_replicas]
        assert len(mine) == self.num_samples
        return iter(mine)
