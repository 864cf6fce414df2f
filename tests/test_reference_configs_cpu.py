"""north_star: "keeps the mmdet model/config API (configs/car_cfg.py ...)".  The reference's OWN config files, unmodified,
load through sassd.config.Config and build through build_detector, and they equal this repo's configs/ key for key except
for the dataset paths.  Skipped when /root/reference is not mounted (the GPU box)."""
import os

import pytest

import sassd  # noqa: F401
from sassd.config import Config
from sassd.detector import build_detector

REF = "/root/reference/configs"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH_KEYS = {"root", "ann_file", "img_prefix", "data_root", "db_info_path", "work_dir", "filename", "load_from",
             "resume_from", "save_to_file"}


def _plain(v):
    if isinstance(v, dict):
        return {k: _plain(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    return v


def _diff(a, b, path=""):
    """paths at which two plain structures differ, ignoring dataset / work-dir path strings"""
    out = []
    if isinstance(a, dict) and isinstance(b, dict):
        for k in sorted(set(a) | set(b)):
            if k in PATH_KEYS:
                continue
            if k not in a or k not in b:
                out.append("%s.%s (only in %s)" % (path, k, "reference" if k in a else "repo"))
            else:
                out += _diff(a[k], b[k], "%s.%s" % (path, k))
    elif isinstance(a, list) and isinstance(b, list):
        if len(a) != len(b):
            out.append("%s (length %d != %d)" % (path, len(a), len(b)))
        else:
            for i, (x, y) in enumerate(zip(a, b)):
                out += _diff(x, y, "%s[%d]" % (path, i))
    elif isinstance(a, str) and isinstance(b, str) and ("/" in a or "/" in b):
        pass                                             # a path inside a list / nested value
    elif a != b:
        out.append("%s: %r != %r" % (path, a, b))
    return out


@pytest.mark.parametrize("name", ["car_cfg.py", "multi_cfg.py"])
def test_reference_config_loads_unmodified_and_equals_ours(name):
    if not os.path.isfile(os.path.join(REF, name)):
        pytest.skip("the reference tree is not mounted here")
    ref = Config.fromfile(os.path.join(REF, name))
    ours = Config.fromfile(os.path.join(ROOT, "configs", name))
    d = _diff(_plain(ref), _plain(ours))
    assert not d, "\n".join(d[:40])
    model = build_detector(ref.model, ref.train_cfg, ref.test_cfg)
    mine = build_detector(ours.model, ours.train_cfg, ours.test_cfg)
    sa, sb = model.state_dict(), mine.state_dict()
    assert list(sa) == list(sb) and all(sa[k].shape == sb[k].shape for k in sa)
    assert sum(p.numel() for p in model.parameters()) == (5339548 if name == "car_cfg.py" else
                                                          sum(p.numel() for p in mine.parameters()))
