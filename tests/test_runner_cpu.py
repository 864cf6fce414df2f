"""The epoch / test loops of sassd.runner with stand-in model, optimizer and loader (the real ones need the GPU and are
exercised by tests/test_gpu_train.py): iteration accounting, scheduler calls, log cadence, checkpoint rotation, frame
sharding and result order of single_test, result files."""
import glob
import logging
import os

import numpy as np
import torch

import sassd  # noqa: F401
from sassd import kitti_common as kc, runner as R, train as T


class _Opt:
    def __init__(self):
        self.lr, self.mom, self.steps, self.zeroed = 0.1, 0.9, 0, 0

    def zero_grad(self):
        self.zeroed += 1

    def step(self):
        self.steps += 1

    def state_dict(self):
        return dict(steps=self.steps)


class _Sched:
    def __init__(self):
        self.calls = []

    def step(self, it):
        self.calls.append(it)


class _Sync:
    n = 0

    def all_reduce_grads(self):
        self.n += 1


class _Model(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.ones(3))

    def forward(self, x=None, img_meta=None, return_loss=True, **kw):
        if return_loss:
            return dict(loss_a=(self.w * x).sum(), loss_b=[self.w.sum() * 0.5], acc=torch.tensor(1.0))
        i = img_meta[0]['sample_idx']
        a = kc.empty_result_anno() if i == 2 else dict(
            name=np.array(['Car']), truncated=np.zeros(1), occluded=np.zeros(1, int), alpha=np.array([0.1 * i]),
            bbox=np.array([[1., 2., 30., 40.]]), dimensions=np.array([[3.9, 1.5, 1.6]]), location=np.array([[1., 1.6, 10. + i]]),
            rotation_y=np.array([0.2]), score=np.array([0.5]), image_idx=np.array([i]))
        return [a]


class _Loader(list):
    sampler = type("S", (), {"epochs": [], "set_epoch": lambda self, e: self.epochs.append(e)})()


def test_train_model_loop(tmp_path, caplog):
    model, opt, sched, sync = _Model(), _Opt(), _Sched(), _Sync()
    loader = _Loader(dict(x=torch.full((3,), float(i))) for i in range(5))
    log = logging.getLogger("sassd-test")
    with caplog.at_level(logging.INFO, logger="sassd-test"):
        it = R.train_model(model, opt, loader, sched, sync, start_epoch=1, total_epochs=5, start_iter=5, rank=0, logger=log,
                           ckpt_save_dir=str(tmp_path), ckpt_save_interval=1, max_ckpt_save_num=2, log_interval=2)
    assert it == 5 + 4 * 5 and opt.steps == 20 and opt.zeroed == 20 and sync.n == 20
    assert sched.calls == list(range(5, 25)) and loader.sampler.epochs == [1, 2, 3, 4]
    lines = [r.getMessage() for r in caplog.records]
    assert len(lines) == 4 * 2 and lines[0].startswith("epoch[2][2/5]: lr: 0.100000, loss_a:") and "acc: 1.0" in lines[0]
    files = sorted(os.path.basename(f) for f in glob.glob(str(tmp_path / "checkpoint_epoch_*.pth")))
    assert files == ["checkpoint_epoch_4.pth", "checkpoint_epoch_5.pth"]
    ck = torch.load(str(tmp_path / "checkpoint_epoch_5.pth"), weights_only=False)
    assert (ck["epoch"], ck["it"]) == (5, 25) and torch.equal(ck["model_state"]["w"], model.w.detach())
    total, terms = T.parse_losses(model(x=torch.ones(3)))
    assert float(total) == 3.0 + 1.5 and set(terms) == {"loss_a", "loss_b", "acc"}


class _DS:
    test_mode, with_label, class_names = True, False, ['Car']

    def __init__(self, n):
        self.sample_ids = list(range(100, 100 + n))

    def __len__(self):
        return len(self.sample_ids)

    def load_frame(self, idx, with_label=True):
        return dict(sample_idx=idx)

    def prepare_test_img(self, idx, frame=None):
        return dict(img_meta=dict(sample_idx=frame['sample_idx']))

    def collate(self, samples, model=None):
        return dict(img_meta=[s['img_meta'] for s in samples], return_loss=False)


def test_single_test_order_and_files(tmp_path):
    ds, model = _DS(5), _Model()
    parts = {r: R.single_test(model, ds, rank=r, world=1 if r == 0 else 2) for r in (0,)}
    out = parts[0]
    assert [len(a['name']) for a in out] == [1, 1, 0, 1, 1] and [int(a['image_idx'][0]) for a in out if len(a['name'])] == [0, 1, 3, 4]
    assert model.class_names if hasattr(model, 'class_names') else True
    R.single_test(model, ds, saveto=str(tmp_path / "res"), class_names=['Car'], rank=0, world=1)
    assert sorted(os.listdir(tmp_path / "res")) == ["000000.txt", "000001.txt", "000003.txt", "000004.txt"]
    back = kc.get_label_annos(str(tmp_path / "res"), [3])[0]
    assert back["name"][0] == "Car" and abs(back["location"][0, 2] - 13.0) < 1e-4 and abs(back["score"][0] - 0.5) < 1e-4
    assert model.class_names == ['Car'] and not model.training


def test_load_params_from_file(tmp_path):
    m = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.BatchNorm1d(3))
    flat = T.FlatParams(m)
    ref = {k: torch.randn_like(v.float()).to(v.dtype) for k, v in m.state_dict().items()}
    state = {'module.' + k: v for k, v in ref.items()}
    state['module.0.bias'] = torch.zeros(5)                       # wrong shape -> skipped
    state['module.extra'] = torch.zeros(1)                        # unknown -> skipped
    torch.save(dict(epoch=3, it=9, model_state=state, optimizer_state=None, version='none'), str(tmp_path / "c.pth"))
    before = m[0].bias.detach().clone()
    loaded, skipped = T.load_params_from_file(m, str(tmp_path / "c.pth"), to_cpu=True)
    assert sorted(skipped) == ['module.0.bias', 'module.extra'] and '0.weight' in loaded and '1.running_var' in loaded
    assert torch.equal(m[0].weight, ref['0.weight']) and torch.equal(m[0].bias, before)
    assert m[0].weight.data_ptr() == flat.data.data_ptr() + 4 * flat.offsets[0]      # still a view of the flat buffer
    import pytest
    with pytest.raises(FileNotFoundError):
        T.load_params_from_file(m, str(tmp_path / "missing.pth"))
