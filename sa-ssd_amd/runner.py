"""The loops around the hot path: `train_one_epoch` / `train_model` (tools/train_utils/__init__.py:36-118) and
`single_test` + KITTI evaluation (tools/test.py:19-99,131-158) -- library functions, not a CLI.

One process per GPU.  Training: every rank iterates its DistributedGroupSampler share (sassd.loader), gradients meet in
ONE all-reduce over the flat buffer per step (sassd.train.GradSync), rank 0 logs and writes checkpoints.  Testing: the
reference walks the val split serially on one GPU; here `single_test` takes this rank's round-robin share of the frames
(no data-path collective) and the per-frame result annotations are gathered on the host in dataset order."""
import glob
import os

import torch

from . import dist as D
from . import kitti_common as kitti
from . import train as T
from .loader import FrameLoader


def train_one_epoch(model, optimizer, train_loader, lr_scheduler, sync, accumulated_iter, train_epoch, rank=0,
                    logger=None, log_interval=20, lr_warmup_scheduler=None):
    """-> accumulated_iter after the epoch.  Loss terms are kept on the device and only read every `log_interval`
    iterations (the reference calls .item() on every term of every iteration, train_utils/__init__.py:8-25)."""
    if hasattr(train_loader, 'sampler') and hasattr(train_loader.sampler, 'set_epoch'):
        train_loader.sampler.set_epoch(train_epoch - 1)
    window = []
    for i, batch in enumerate(train_loader):
        warm = lr_warmup_scheduler is not None and accumulated_iter <= lr_warmup_scheduler.T_max
        _, terms = T.train_one_iter(model, optimizer, lr_warmup_scheduler if warm else lr_scheduler, sync, batch,
                                    accumulated_iter)
        accumulated_iter += 1
        window.append(terms)
        if (i + 1) % log_interval == 0:
            if rank == 0 and logger is not None:
                mean = {k: float(torch.stack([w[k] for w in window]).mean()) for k in window[0]}
                logger.info('epoch[%d][%d/%d]: lr: %f, ' % (train_epoch, i + 1, len(train_loader), float(optimizer.lr))
                            + ', '.join('%s: %f' % kv for kv in mean.items()))
            window = []
    return accumulated_iter


def train_model(model, optimizer, train_loader, lr_scheduler, sync, start_epoch, total_epochs, start_iter, rank=0,
                logger=None, ckpt_save_dir=None, lr_warmup_scheduler=None, ckpt_save_interval=1, max_ckpt_save_num=50,
                log_interval=20):
    """Epoch loop with `checkpoint_epoch_%d.pth` files, oldest removed beyond max_ckpt_save_num (rank 0 only)."""
    it = start_iter
    for cur_epoch in range(start_epoch, total_epochs):
        trained = cur_epoch + 1
        it = train_one_epoch(model, optimizer, train_loader, lr_scheduler, sync, it, trained, rank, logger, log_interval,
                             lr_warmup_scheduler)
        if ckpt_save_dir is not None and trained % ckpt_save_interval == 0 and rank == 0:
            os.makedirs(ckpt_save_dir, exist_ok=True)
            old = sorted(glob.glob(os.path.join(ckpt_save_dir, 'checkpoint_epoch_*.pth')), key=os.path.getmtime)
            for f in old[:max(0, len(old) - max_ckpt_save_num + 1)]:
                os.remove(f)
            T.save_checkpoint(T.checkpoint_state(model, optimizer, trained, it),
                              os.path.join(ckpt_save_dir, 'checkpoint_epoch_%d' % trained))
    return it


def single_test(model, dataset, saveto=None, class_names=None, workers=2, rank=None, world=None):
    """Run the detector over `dataset` (test mode) -> list of KITTI result annotations in dataset order on every rank
    (this rank's frames are computed here, the others' gathered from their ranks).  `saveto`: also write result files."""
    if rank is None or world is None:
        rank, _, world = D.env_world() if D.dist.is_initialized() else (0, 0, 1)
    if class_names is not None:
        setattr(model, 'class_names', class_names)
    model.eval()
    mine = D.frame_shard(len(dataset), rank, world)
    annos = []
    with torch.no_grad():
        for batch in FrameLoader(dataset, 1, sampler=mine, num_workers=workers):
            annos += model(**batch)
    merged = [None] * len(dataset)
    for part_rank, part in enumerate(D.gather_results(annos) if world > 1 else [annos]):
        for idx, anno in zip(D.frame_shard(len(dataset), part_rank, world), part):
            merged[idx] = anno
    if saveto is not None and rank == 0:
        kitti.write_label_annos(merged, saveto)
    return merged


def evaluate(dataset, outputs, class_names=None):
    """tools/test.py:155-158: the official KITTI report of `outputs` against the dataset's label files."""
    from .kitti_eval import get_official_eval_result
    gt_annos = kitti.get_label_annos(dataset.label_prefix, dataset.sample_ids)
    return get_official_eval_result(gt_annos, outputs, current_classes=class_names or dataset.class_names)
