"""Which Python lines issue the training step's torch-level calls (dtype conversions, copies, fills, cats, arithmetic)?
A TorchFunctionMode around the training bench loop counts every torch API call by its innermost repo frame:
    python tools/trace_train_launch_sites.py [--steps 6]
(the autograd engine's own backward nodes -- SliceBackward0, DivBackward0 ... -- do not pass through here: they mirror the
forward slicing / arithmetic sites that do)."""
import collections
import os
import sys

import torch
from torch.overrides import TorchFunctionMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
steps = 6
if "--steps" in sys.argv:
    steps = int(sys.argv[sys.argv.index("--steps") + 1])
sys.argv = ["bench.py", "--mode", "train", "--steps", str(steps), "--warmup", "3"]
import bench  # noqa: E402

counts = collections.Counter()
SKIP = ("size", "dim", "shape", "__get__", "numel", "stride", "is_contiguous", "data_ptr", "element_size", "storage_offset",
        "view", "reshape", "permute", "transpose", "unsqueeze", "squeeze", "detach", "requires_grad_", "_make_subclass",
        "is_floating_point", "t", "expand", "unbind", "split", "chunk", "narrow", "select", "__len__", "__bool__",
        "tolist", "__hash__", "__format__", "__repr__", "is_cuda", "device", "dtype", "grad", "__set__", "record_stream",
        "apply", "backward", "__iter__", "ndim", "view_as", "flatten", "contiguous_format")


class Tracer(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        name = getattr(func, "__name__", str(func))
        if name not in SKIP:
            f = sys._getframe(1)
            site = "?"
            while f is not None:
                fn = f.f_code.co_filename
                if ("sa-ssd_amd" in fn or "sassd" in fn or fn.endswith("bench.py")) and "tools/" not in fn:
                    site = "%s:%d" % (fn[fn.rfind("sa-ssd_amd"):] if "sa-ssd_amd" in fn else os.path.basename(fn), f.f_lineno)
                    break
                f = f.f_back
            counts[(name, site)] += 1
        return func(*args, **(kwargs or {}))


from sassd import train as T  # noqa: E402

orig_iter = T.train_one_iter
calls = [0]


def traced_iter(*a, **k):
    calls[0] += 1
    if calls[0] <= 3:                        # warm-up steps: lazily built caches
        return orig_iter(*a, **k)
    with Tracer():
        return orig_iter(*a, **k)


T.train_one_iter = traced_iter
bench.main()
total = max(calls[0] - 3, 1)
print("torch API calls per call site, per training step (%d steps traced; data preparation of the next batch included):" % total)
for (name, site), n in counts.most_common(160):
    if n >= total:
        print("%7.1f  %-22s %s" % (n / total, name, site))
