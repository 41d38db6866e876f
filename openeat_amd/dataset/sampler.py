"""Which pre-formed batch goes to which rank at which step.

The reference hands its list of batches to torch.utils.data.DistributedSampler (train_ddp.py:131-134): a seeded
permutation of batch indices, padded to a multiple of the world size, rank r taking every world-th index.  With
length-sorted 'dynamic' batches that puts a batch of many short utterances beside a batch of few long ones in the same
step: the ranks' step times differ and the gradient all-reduce waits for the slowest.

mode="reference": exactly DistributedSampler's indices (same generator, same padding) - checked index by index against
    the torch class in tests/test_host_logic.py.
mode="balanced" : MI355X-first.  Batches are ranked by cost (padded frames), cut into groups of `world` neighbours, and
    a step takes one group: all ranks of a step get batches of nearly equal cost.  The seeded permutation shuffles the
    ORDER of groups and rotates the assignment inside a group, so over epochs every rank sees every kind of batch.
    Every batch is used exactly once per epoch (the last group is filled up with batches of the nearest cost, as
    DistributedSampler pads with repeats).
Pure index arithmetic on the host: no collective, every rank computes the same plan from (seed, epoch)."""
import math
from typing import Iterator, List, Optional, Sequence

import torch


class DistributedBatchSampler(torch.utils.data.Sampler):
    def __init__(self, num_batches: int, num_replicas: int, rank: int, shuffle: bool = True, seed: int = 0,
                 mode: str = "reference", costs: Optional[Sequence[float]] = None):
        assert mode in ("reference", "balanced")
        assert 0 <= rank < num_replicas
        if mode == "balanced":
            assert costs is not None and len(costs) == num_batches, "balanced mode needs one cost per batch"
        self.n, self.world, self.rank, self.shuffle, self.seed, self.mode = num_batches, num_replicas, rank, shuffle, seed, mode
        self.costs = None if costs is None else [float(c) for c in costs]
        self.epoch = 0
        self.num_samples = math.ceil(num_batches / num_replicas)
        self.total_size = self.num_samples * num_replicas

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def __len__(self):
        return self.num_samples

    def plan(self) -> List[List[int]]:
        """[step][rank] -> batch index, identical on every rank."""
        g = torch.Generator()
        g.manual_seed(self.seed + self.epoch)
        if self.mode == "reference":
            idx = torch.randperm(self.n, generator=g).tolist() if self.shuffle else list(range(self.n))
            pad = self.total_size - len(idx)
            if pad > 0:
                idx += (idx * math.ceil(pad / len(idx)))[:pad] if pad > len(idx) else idx[:pad]
            return [idx[s * self.world:(s + 1) * self.world] for s in range(self.num_samples)]
        order = sorted(range(self.n), key=lambda i: (self.costs[i], i))
        pad = self.total_size - self.n
        if pad > 0:                          # fill the last group with its own nearest neighbours
            order += order[-self.world:][:pad] if self.n >= self.world else (order * self.world)[:pad]
        groups = [order[s * self.world:(s + 1) * self.world] for s in range(self.num_samples)]
        if self.shuffle:
            perm = torch.randperm(len(groups), generator=g).tolist()
            rot = torch.randint(0, self.world, (len(groups),), generator=g).tolist()
            groups = [groups[p][r:] + groups[p][:r] for p, r in zip(perm, rot)]
        return groups

    def __iter__(self) -> Iterator[int]:
        return iter([step[self.rank] for step in self.plan()])


def step_imbalance(plan: List[List[int]], costs: Sequence[float]) -> float:
    """Mean over steps of (max over ranks - mean over ranks) / max over ranks: the share of a step the average rank
    spends waiting at the all-reduce if step time is proportional to cost."""
    tot = 0.0
    for step in plan:
        c = [costs[i] for i in step]
        tot += (max(c) - sum(c) / len(c)) / max(c)
    return tot / max(len(plan), 1)
