"""Synthetic KITTI-range / Waymo-scale LiDAR frames (measurement contract).

These generators define the bench/parity inputs; RNG draw order matters for the quoted voxel counts
(seed 0: K21 -> 16111 voxels, K17 -> 13435, waymo_synth -> 79302).  Pure numpy, no reference code.
"""
import numpy as np

KITTI_RANGE = (0.0, -40.0, -3.0, 70.4, 40.0, 1.0)     # configs/car_cfg.py generator.point_cloud_range
KITTI_VOXEL = (0.05, 0.05, 0.1)
WAYMO_RANGE = (-75.2, -75.2, -2.0, 75.2, 75.2, 4.0)
WAYMO_VOXEL = (0.1, 0.1, 0.15)


def _ray_cloud(seed, elev_deg, az_rad, r_ob, r_max, crop):
    rng = np.random.default_rng(seed)
    e = np.deg2rad(np.linspace(elev_deg[0], elev_deg[1], 64))
    E, A = np.meshgrid(e, az_rad, indexing="ij")
    E = E.ravel()
    A = A.ravel()
    n = E.size
    rg = np.where(E < 0, 1.73 / np.maximum(np.sin(-E), 1e-9), np.inf)   # ground plane z = -1.73
    u = rng.random(n)
    rob = rng.uniform(r_ob[0], r_ob[1], n)
    r = np.where(E < 0, np.where((u < 0.25) & (rob < rg), rob, rg), np.where(u < 0.35, rob, np.inf))
    r = r + rng.normal(0.0, 0.02, n)
    keep = np.isfinite(r) & (r < r_max)
    r, E, A = r[keep], E[keep], A[keep]
    x = r * np.cos(E) * np.cos(A)
    y = r * np.cos(E) * np.sin(A)
    z = r * np.sin(E)
    inten = rng.random(r.size)
    pts = np.stack([x, y, z, inten], 1).astype(np.float32)
    lo = np.asarray(crop[:3], np.float32)
    hi = np.asarray(crop[3:], np.float32)
    m = np.all((pts[:, :3] >= lo) & (pts[:, :3] < hi), axis=1)
    pts = pts[m]
    rng.shuffle(pts)
    return np.ascontiguousarray(pts)


def lidar64(seed=0, n_az=469, fov=40.5, elev=(2.0, -24.8), r_ob=(5.0, 70.0), r_max=80.0,
            crop=KITTI_RANGE):
    """KITTI-like 64-beam frontal cloud. seed 0 -> 27124 rows; K21 = [:21500], K17 = [:17000]."""
    a = np.deg2rad(np.linspace(-fov, fov, n_az))
    return _ray_cloud(seed, elev, a, r_ob, r_max, crop)


def waymo_synth(seed=0):
    """360-degree 64-beam cloud, 3300 azimuths. seed 0 -> 184569 rows; use [:180000]."""
    a = np.deg2rad(np.linspace(-180.0, 180.0, 3300, endpoint=False))
    crop = (-75.2, -75.2, -2.0, 75.2, 75.2, 4.0)
    return _ray_cloud(seed, (2.4, -17.6), a, (5.0, 75.0), 110.0, crop)


def k21(seed=0):
    return lidar64(seed)[:21500]


def k17(seed=0):
    return lidar64(seed)[:17000]


# ----------------------------------------------------------------------------------------------------------------
# Synthetic WEIGHTS and named workloads for bench.py / smoke() / tests (BASELINE.json configs[1], [3], [4]).
# Product-side helpers: nothing here touches the CPU oracle.
# ----------------------------------------------------------------------------------------------------------------
def randomize_detector(model, seed=0, cls_bias=-2.0, sparse_fan_div=3):
    """Seeded weights + randomised BN running stats (so BN folding is exercised) + a negative cls bias so that
    ~10^2 anchors pass the 0.1 guided-anchor threshold.  `sparse_fan_div`: the 27-offset sparse kernels
    see ~1/3 of their taps on KITTI-like clouds (He gain over fan_in / 3 keeps activations O(1) there); dense
    Waymo-scale clouds fill most taps and need the plain fan_in (3 -> 1), else activations grow ~1.7x per layer."""
    import torch
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() >= 2:
                fan_in = p[0].numel() if p.dim() == 4 else int(np.prod(p.shape[:-1]))
                if p.dim() == 5:
                    fan_in = int(np.prod(p.shape[:4])) // sparse_fan_div          # sparse kernels are mostly empty
                p.copy_(torch.randn(p.shape, generator=g) * (2.0 / max(fan_in, 1)) ** 0.5)
            elif name.endswith("bias"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
            else:
                p.copy_(torch.rand(p.shape, generator=g) * 0.5 + 0.75)
        for name, b in model.named_buffers():
            if name.endswith("running_mean"):
                b.copy_(torch.randn(b.shape, generator=g) * 0.1)
            elif name.endswith("running_var"):
                b.copy_(torch.rand(b.shape, generator=g) + 0.5)
        model.rpn_head.conv_cls.bias.add_(cls_bias)
        # keep decoded boxes close to their anchors (|t| < ~0.5: sizes exp(t) * anchor stay car-sized, as with trained
        # weights; unbounded random regressions decode to 10-100 m boxes whose absolute fp32 error means nothing)
        model.rpn_head.conv_box.weight.mul_(0.02)
        model.rpn_head.conv_box.bias.mul_(0.5)
    return model


_ANCHOR_SIZES = dict(Car=[1.6, 3.9, 1.56], Pedestrian=[0.6, 0.8, 1.73], Cyclist=[0.6, 1.76, 1.73])


def workload(name):
    """Named measurement workloads -> dict(cfg file, class names, model-config overrides, InferencePlan kwargs,
    anchors [A,7], anchors_bv [A,4], frame(seed) -> points, points_cap, default batch, description)."""
    from . import anchors as A
    if name in ("car", "multi"):
        names = ["Car"] if name == "car" else ["Car", "Pedestrian", "Cyclist"]
        an = np.concatenate([A.AnchorGeneratorStride(sizes=_ANCHOR_SIZES[n], anchor_strides=[.4, .4, 1.],
                                                     anchor_offsets=[.2, -39.8, -1.78], rotations=[0, 1.57])(
            [1, 200, 176]).reshape(-1, 7) for n in names], 0)
        w = dict(cfg="configs/%s_cfg.py" % name, class_names=names, overrides=None, frame=k21, points_cap=21504,
                 plan=dict(num_class=len(names), cap_k=4096, cap_d=512 if name == "car" else 1024),
                 cal=dict(voxel_size=KITTI_VOXEL, pc_range=KITTI_RANGE, max_points=5, max_voxels=20000,
                          sparse_shape=(40, 1600, 1408), grid_xyz=(1408, 1600, 40)),
                 batch=1 if name == "car" else 8,
                 desc="configs/%s_cfg.py inference, fp32, synthetic lidar64 K21 frames (21500 pts -> ~16k voxels)" % name)
    elif name == "waymo":
        an = A.AnchorGeneratorStride(sizes=_ANCHOR_SIZES["Car"], anchor_strides=[.8, .8, 1.],
                                     anchor_offsets=[-74.8, -74.8, -1.0], rotations=[0, 1.57])([1, 188, 188]).reshape(-1, 7)
        w = dict(cfg="configs/car_cfg.py", class_names=["Car"],
                 overrides=dict(neck=dict(output_shape=[40, 1504, 1504], aux_offset=WAYMO_RANGE[:3],
                                          aux_voxel_size=WAYMO_VOXEL),
                                extra_head=dict(grid_offsets=(75.2, 75.2), featmap_stride=0.8)),
                 frame=lambda seed=0: waymo_synth(seed)[:180000], points_cap=180000,
                 plan=dict(num_class=1, voxel_size=WAYMO_VOXEL, point_cloud_range=WAYMO_RANGE, max_voxels=150000,
                           sparse_shape=(40, 1504, 1504), grid_offsets=(75.2, 75.2), featmap_stride=0.8, cap_k=4096,
                           cap_d=2048),
                 cal=dict(voxel_size=WAYMO_VOXEL, pc_range=WAYMO_RANGE, max_points=5, max_voxels=150000,
                          sparse_shape=(40, 1504, 1504), grid_xyz=(1504, 1504, 40)),
                 batch=4,
                 desc="Waymo-scale synthetic inference (car head), fp32: 180000 pts/frame, 0.1x0.1x0.15 m voxels "
                      "(grid 40x1504x1504, ~79k active), BEV 188x188")
    else:
        raise KeyError(name)
    w["name"] = name
    w["anchors"] = np.ascontiguousarray(an, np.float32)
    w["anchors_bv"] = A.rbbox2d_to_near_bbox(an[:, [0, 1, 3, 4, 6]]).astype(np.float32)
    return w


def build_detector_for(w, seed=0, train=False, cls_bias=-2.0):  # noqa: C901
    """Random-init SA-SSD for workload `w` (reference config files loaded unmodified, overrides applied on top)."""
    import os
    from .config import Config
    from .detector import build_detector
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, w["cfg"]))
    mcfg = dict(cfg.model)
    for k, v in (w["overrides"] or {}).items():
        mcfg[k] = dict(mcfg[k], **v)
    model = build_detector(mcfg, cfg.train_cfg, cfg.test_cfg)
    model = model if train else model.eval()
    return randomize_detector(model, seed, cls_bias=cls_bias, sparse_fan_div=1 if w["name"] == "waymo" else 3), cfg


def calibrate_cls_head_on_device(model, w, dev, cloud, target_count=100, target_std=0.45):
    """Rescale rpn_head.conv_cls so that about `target_count` masked anchors pass sigmoid > 0.1 on `cloud` (SURVEY.md
    8d asks for K ~ 10^2-10^3 candidates, raw random weights give tens of thousands).  One run of the HIP pipeline."""
    import torch
    from .pipeline import InferencePlan
    plan = InferencePlan(model.state_dict(), batch_size=1, anchors=w["anchors"], anchors_bv=w["anchors_bv"], device=dev,
                         **w["plan"])
    plan.run_from_points([torch.from_numpy(np.ascontiguousarray(cloud)).to(dev)])
    torch.cuda.synchronize()
    hw, ncls = plan.H * plan.W, plan.ncls
    cls = plan.head_out.view(-1)[plan.n_box * hw:(plan.n_box + plan.n_cls) * hw]
    # channel = group*(A*ncls) + a*ncls + c ; anchor index = ((group*H + h)*W + w)*A + a   (ssd_rotate_head.py:228-233)
    lg = cls.view(ncls, plan.A, ncls, hw).permute(0, 3, 1, 2).reshape(-1, ncls)
    lg = lg[plan.mask[0].bool()].max(-1)[0].double().cpu()
    b_old = model.rpn_head.conv_cls.bias.detach().double().cpu()
    sc = target_std / max(float(lg.std()), 1e-6)
    q = float(torch.quantile((lg - b_old.mean()) * sc, 1.0 - min(0.5, target_count / max(lg.numel(), 1))))
    with torch.no_grad():
        model.rpn_head.conv_cls.weight.mul_(sc)
        model.rpn_head.conv_cls.bias.copy_(((b_old - b_old.mean()) * sc + (float(np.log(0.1 / 0.9)) - q)).float())
    return model
