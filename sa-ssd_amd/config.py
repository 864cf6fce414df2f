"""mmcv-free stand-ins for the two mmcv pieces the reference's construction path needs:
`mmcv.Config.fromfile` (tools/test.py:128) and `mmcv.runner.obj_from_dict` (mmdet/models/builder.py:13-16).
A config is a plain Python file; nested dicts become attribute dicts; `type` strings name classes."""
import importlib.util
import os


class ConfigDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(v):
    if isinstance(v, dict):
        return ConfigDict({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, (list, tuple)):
        return type(v)(_wrap(x) for x in v)
    return v


class Config(ConfigDict):
    @staticmethod
    def fromfile(path):
        path = os.path.abspath(path)
        spec = importlib.util.spec_from_file_location("_sassd_cfg", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        d = {k: v for k, v in vars(mod).items() if not k.startswith("_") and not isinstance(v, type(os))}
        cfg = Config(_wrap(d))
        dict.__setitem__(cfg, "filename", path)
        return cfg


def obj_from_dict(info, parent=None, default_args=None):
    """Instantiate `info['type']` (a class, or the name of an attribute of `parent`) with the remaining keys."""
    args = dict(info)
    t = args.pop("type")
    if isinstance(t, str):
        t = getattr(parent, t)
    if default_args:
        for k, v in default_args.items():
            args.setdefault(k, v)
    return t(**args)
