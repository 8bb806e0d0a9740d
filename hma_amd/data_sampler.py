"""Temperature-sampled multi-dataset batch sampler (SURVEY (f) row 3).

Mirror of `MultiTaskBatchSampler`, external/data_sampler.py:175-313, as `train_multi.py:928-932` builds it: one
domain per batch, domains drawn from `w_i ∝ (n_i / Σn)^(1/temperature)`, samples drawn with replacement from that
rank's shard of the (per-epoch shuffled) dataset.  Host logic only.  It consumes the seeded `torch.Generator` in the
reference's order (one `randperm` per dataset, optional reseed with `seed + epoch + rank`, one `multinomial`, one
`randint` per batch), so for the same arguments it yields the reference's batches exactly
(tests/test_data_cpu.py against tests/golden/g10_data.json).
"""
from __future__ import annotations

from typing import Iterator, List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist
from torch.utils.data import Sampler


class MultiTaskBatchSampler(Sampler):
    def __init__(self, dataset_sizes: List[int], batch_size: int, temperature: float, dataset_groups: Sequence = (),
                 num_replicas: Optional[int] = 1, rank: Optional[int] = 0, seed: int = 0, shuffle: bool = True,
                 shuffle_task: bool = True) -> None:
        if num_replicas is None or rank is None:
            if not (dist.is_available() and dist.is_initialized()):
                raise RuntimeError("num_replicas / rank left to torch.distributed, which is not initialised")
            num_replicas = dist.get_world_size() if num_replicas is None else num_replicas
            rank = dist.get_rank() if rank is None else rank
        if not 0 <= rank < num_replicas:
            raise ValueError(f"Invalid rank {rank}, rank should be in the interval [0, {num_replicas - 1}]")
        self.dataset_groups = list(dataset_groups)
        self.num_replicas, self.rank = num_replicas, rank
        self.batch_size, self.temperature = batch_size, temperature
        self.dataset_sizes = list(dataset_sizes)
        self.seed, self.epoch = seed, 0
        self.shuffle, self.shuffle_task = shuffle, shuffle_task
        # the tail that does not divide over the ranks is dropped (data_sampler.py:229-234)
        self.rank_dataset_sizes = [n // num_replicas for n in self.dataset_sizes]
        self.total_sizes = [(n // num_replicas) * num_replicas for n in self.dataset_sizes]
        self.dataset_offsets = torch.cumsum(torch.LongTensor([0] + self.dataset_sizes), 0)
        self.num_batches_per_epoch = int((np.sum(self.dataset_sizes) + batch_size - 1) // batch_size // num_replicas)

    def generate_tasks_distribution(self) -> torch.Tensor:
        def temp(sizes):
            tot = sum(sizes)
            w = np.array([(n / tot) ** (1.0 / self.temperature) for n in sizes])
            return w / np.sum(w)

        if self.dataset_groups:  # normalise inside each [lo, hi) group, groups weighted equally (data_sampler.py:249-259)
            parts = [temp([self.dataset_sizes[i] for i in range(lo, hi)]) / len(self.dataset_groups)
                     for lo, hi in self.dataset_groups]
            weights = np.concatenate(parts)
        else:
            weights = temp(self.dataset_sizes)
        return torch.as_tensor(weights, dtype=torch.double)

    def __iter__(self) -> Iterator[List[int]]:
        gen = torch.Generator()
        gen.manual_seed(self.seed + self.epoch)
        shard = []
        for n, total in zip(self.dataset_sizes, self.total_sizes):
            order = torch.randperm(n, generator=gen) if self.shuffle else torch.arange(n)
            shard.append(order[self.rank:total:self.num_replicas])
        self.rank_indices = [s.tolist() for s in shard]
        weights = self.generate_tasks_distribution()
        if self.shuffle_task:  # ranks draw different domain sequences (data_sampler.py:293-295)
            gen.manual_seed(self.seed + self.epoch + self.rank)
        tasks = torch.multinomial(weights, self.num_batches_per_epoch, replacement=True, generator=gen)
        for task in tasks.tolist():
            pick = torch.randint(low=0, high=self.rank_dataset_sizes[task], size=(self.batch_size,), generator=gen)
            yield (self.dataset_offsets[task] + shard[task][pick]).tolist()

    def __len__(self) -> int:
        return self.num_batches_per_epoch

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch
