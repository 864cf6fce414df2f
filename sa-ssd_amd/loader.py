"""Batch loop for one GPU per process: the samplers and `build_dataloader` of
mmdet/datasets/loader/{sampler,build_loader}.py, without torch's DataLoader workers.

The reference starts `workers_per_gpu` processes per GPU that each run the numba augmentation / voxelisation on the CPU
and ship collated voxel tensors through shared memory and a pinned copy.  Here the only host work per frame is reading
three small files (`KittiLiDAR.load_frame`), done `workers_per_gpu` frames ahead by a thread pool (file reads release the
GIL); augmentation, voxelisation, anchor masks and rulebooks run on the GPU in the consuming process
(`prepare_*_img(frame=...)` + `collate`), so nothing crosses a process boundary and nothing is copied twice.

  GroupSampler / DistributedGroupSampler   same index sequences as the reference's (sampler.py:11-131): per-epoch
                                           torch.randperm with generator seed = epoch, padding to a multiple of
                                           samples_per_gpu x num_replicas, batch-wise permutation, contiguous rank slice
  build_dataloader(dataset, imgs_per_gpu, workers_per_gpu, num_gpus=1, dist=True, **kw) -> FrameLoader"""
import math
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch
import torch.distributed as tdist


class GroupSampler:
    def __init__(self, dataset, samples_per_gpu=1):
        self.dataset, self.samples_per_gpu = dataset, samples_per_gpu
        self.flag = dataset.flag.astype(np.int64)
        self.group_sizes = np.bincount(self.flag)
        self.num_samples = sum(int(np.ceil(s / samples_per_gpu)) * samples_per_gpu for s in self.group_sizes)

    def __iter__(self):
        spg, groups = self.samples_per_gpu, []
        for g, size in enumerate(self.group_sizes):
            if size == 0:
                continue
            idx = np.where(self.flag == g)[0]
            np.random.shuffle(idx)
            pad = int(np.ceil(size / spg)) * spg - len(idx)
            groups.append(np.concatenate([idx, idx[:pad]]))
        flat = np.concatenate(groups)
        order = np.random.permutation(range(len(flat) // spg))
        return iter(torch.from_numpy(np.concatenate([flat[b * spg:(b + 1) * spg] for b in order])).long())

    def __len__(self):
        return self.num_samples


class DistributedGroupSampler:
    def __init__(self, dataset, samples_per_gpu=1, num_replicas=None, rank=None):
        self.dataset, self.samples_per_gpu = dataset, samples_per_gpu
        self.num_replicas = tdist.get_world_size() if num_replicas is None else num_replicas
        self.rank = tdist.get_rank() if rank is None else rank
        self.epoch = 0
        self.flag = dataset.flag
        self.group_sizes = np.bincount(self.flag)
        per = samples_per_gpu * self.num_replicas
        self.num_samples = sum(int(math.ceil(s / per)) * samples_per_gpu for s in self.group_sizes)
        self.total_size = self.num_samples * self.num_replicas

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.epoch)                        # every rank draws the same permutation, then takes its slice
        spg, per, flat = self.samples_per_gpu, self.samples_per_gpu * self.num_replicas, []
        for grp, size in enumerate(self.group_sizes):
            if size > 0:
                idx = np.where(self.flag == grp)[0][torch.randperm(int(size), generator=g).numpy()].tolist()
                flat += idx + idx[:int(math.ceil(size / per)) * per - len(idx)]
        order = torch.randperm(len(flat) // spg, generator=g).tolist()
        flat = [flat[j] for b in order for j in range(b * spg, (b + 1) * spg)]
        lo = self.num_samples * self.rank
        return iter(flat[lo:lo + self.num_samples])

    def __len__(self):
        return self.num_samples

    def set_epoch(self, epoch):
        self.epoch = epoch


class FrameLoader:
    """Iterates collated, device-resident batches of `dataset` (a sassd KittiLiDAR).  `len()` = batches per epoch."""

    def __init__(self, dataset, batch_size, sampler=None, num_workers=2, model=None, drop_last=False):
        self.dataset, self.batch_size, self.sampler = dataset, batch_size, sampler
        self.num_workers, self.model, self.drop_last = max(1, int(num_workers)), model, drop_last

    def _indices(self):
        return [int(i) for i in (self.sampler if self.sampler is not None else range(len(self.dataset)))]

    def __len__(self):
        n = len(self.sampler) if self.sampler is not None else len(self.dataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def _prepare(self, idx, frame):
        ds = self.dataset
        if ds.test_mode:
            return ds.prepare_test_img(idx, frame=frame)
        data = ds.prepare_train_img(idx, frame=frame)
        return data if data is not None else ds[idx]             # no object in range: draw another frame, like the reference

    def __iter__(self):
        ds, order = self.dataset, self._indices()
        want_label = (not ds.test_mode) or ds.with_label
        with ThreadPoolExecutor(self.num_workers) as pool:
            ahead = self.num_workers + self.batch_size
            pending = [pool.submit(ds.load_frame, i, want_label) for i in order[:ahead]]
            nxt, samples = len(pending), []
            for pos, idx in enumerate(order):
                frame = pending[pos].result()
                pending[pos] = None
                if nxt < len(order):
                    pending.append(pool.submit(ds.load_frame, order[nxt], want_label))
                    nxt += 1
                samples.append(self._prepare(idx, frame))
                if len(samples) == self.batch_size:
                    yield ds.collate(samples, model=self.model)
                    samples = []
            if samples and not self.drop_last:
                yield ds.collate(samples, model=self.model)


def build_dataloader(dataset, imgs_per_gpu, workers_per_gpu, num_gpus=1, dist=True, model=None, **kwargs):
    """Same arguments as mmdet/datasets/loader/build_loader.py:15-45.  dist=True: this process's share of every epoch
    (DistributedGroupSampler over the process group, or rank 0 of 1 without one); dist=False: GroupSampler, or file order
    with shuffle=False (the test loop, tools/test.py)."""
    if dist:
        on = tdist.is_available() and tdist.is_initialized()
        sampler = DistributedGroupSampler(dataset, imgs_per_gpu, tdist.get_world_size() if on else 1,
                                          tdist.get_rank() if on else 0)
    elif not kwargs.get('shuffle', True) or dataset.test_mode:
        sampler = None
    else:
        sampler = GroupSampler(dataset, imgs_per_gpu)
    return FrameLoader(dataset, imgs_per_gpu * (1 if dist else num_gpus), sampler, workers_per_gpu, model=model)
