# SA-SSD single-class (Car) configuration: same keys / values as the reference's configs/car_cfg.py, so that
# Config.fromfile(...) + build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg) behave identically.
import os as _os
import sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
from _sassd_base import make as _make  # noqa: E402
globals().update(_make(['Car']))
