"""Samplers and the batch loop (mmdet/datasets/loader/*) against index sequences produced by the reference's own
sampler.py (tests/golden/make_golden_sampler.py), plus the prefetching FrameLoader over the tiny prepared KITTI tree
(device arithmetic via the CPU harness; `collate` itself -- voxelizer, anchor mask -- is covered by the GPU tests)."""
import os
import types

import numpy as np
import torch

import sassd  # noqa: F401
from sassd import loader as L

import harness
import test_create_data_cpu as TC

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [(7, 2, 2), (37, 2, 1), (37, 3, 4), (3712, 2, 8), (100, 1, 3)]        # as in golden/make_golden_sampler.py


def test_sampler_sequences():
    R = np.load(os.path.join(HERE, "golden", "sampler_ref.npz"))
    for n, spg, world in CASES:
        ds = types.SimpleNamespace(flag=np.ones(n, dtype=np.uint8))
        for epoch in (0, 1, 7):
            seen = []
            for rank in range(world):
                s = L.DistributedGroupSampler(ds, spg, world, rank)
                s.set_epoch(epoch)
                got = np.array(list(s), dtype=np.int64)
                assert np.array_equal(got, R["dist_%d_%d_%d_e%d_r%d" % (n, spg, world, epoch, rank)]), (n, spg, world, rank)
                assert len(s) == int(R["dist_%d_%d_%d_len" % (n, spg, world)]) == len(got)
                seen.append(got)
            allidx = np.concatenate(seen)
            assert set(allidx.tolist()) == set(range(n)) and len(allidx) - n < spg * world      # complete, padded < 1 round
        np.random.seed(n)
        g = L.GroupSampler(ds, spg)
        assert np.array_equal(np.array([int(i) for i in g]), R["group_%d_%d" % (n, spg)])
        assert np.array_equal(np.array([int(i) for i in g]), R["group_%d_%d_again" % (n, spg)])


def test_frame_loader(tmp_path, monkeypatch):
    harness.patch(monkeypatch)
    from sassd.kitti_dataset import get_dataset
    root = str(tmp_path)
    TC.run_preparation(root, torch.device("cpu"))
    base = dict(type='KittiLiDAR', root=root + '/training/', img_prefix=None, with_point=True, class_names=['Car', 'Pedestrian'],
                generator=dict(type='VoxelGenerator', voxel_size=[0.05, 0.05, 0.1],
                               point_cloud_range=[0, -40., -3., 70.4, 40., 1.], max_num_points=5, max_voxels=20000),
                anchor_generator=dict(Car=dict(type='AnchorGeneratorStride', sizes=[1.6, 3.9, 1.56],
                                               anchor_strides=[0.4, 0.4, 1.0], anchor_offsets=[0.2, -39.8, -1.78],
                                               rotations=[0, 1.57])), anchor_area_threshold=1, out_size_factor=8)
    aug = dict(type='PointAugmentor', root_path=root + '/', info_path=root + '/kitti_dbinfos_train.pkl',
               sample_classes=['Van', 'Pedestrian'], min_num_points=[2, 2], sample_max_num=[4, 3], removed_difficulties=[-1],
               global_rot_range=[-0.78, 0.78], gt_rot_range=[-0.78, 0.78], center_noise_std=[1., 1., .5],
               scale_range=[0.95, 1.05])
    np.random.seed(9)
    ds = get_dataset(dict(base, ann_file=root + '/ImageSets/trainval.txt', with_label=True, augmentor=aug, test_mode=False),
                     device="cpu")
    monkeypatch.setattr(ds, "collate", lambda samples, model=None: samples)      # (the real collate needs the GPU)
    for world in (1, 2):
        batches = {}
        for rank in range(world):
            ld = L.FrameLoader(ds, 2, L.DistributedGroupSampler(ds, 2, world, rank), num_workers=3)
            batches[rank] = list(ld)
            assert len(batches[rank]) == len(ld) and all(len(b) == 2 for b in batches[rank])
            for b in batches[rank]:
                for s in b:
                    assert s['points'].shape[1] == 4 and len(s['gt_bboxes']) == len(s['gt_labels']) > 0
        ids = sorted(s['img_meta']['sample_idx'] for r in batches.values() for b in r for s in b)
        # frame 3 has no labels: whenever the augmentor pastes nothing in range it is replaced by another frame
        assert set(ids) - {3} == {0, 1, 2, 5} and len(ids) == -(-5 // (2 * world)) * 2 * world
    ld = L.build_dataloader(ds, 2, 2, dist=True)                               # no process group: rank 0 of 1
    assert isinstance(ld.sampler, L.DistributedGroupSampler) and ld.sampler.num_replicas == 1 and len(ld) == 3
    dv = get_dataset(dict(base, ann_file=root + '/ImageSets/val.txt', with_label=False, augmentor=None, test_mode=True),
                     device="cpu")
    monkeypatch.setattr(dv, "collate", lambda samples, model=None: samples)
    tl = L.build_dataloader(dv, 1, 2, num_gpus=1, dist=False, shuffle=False)
    assert tl.sampler is None and [b[0]['img_meta']['sample_idx'] for b in tl] == [2, 5]
