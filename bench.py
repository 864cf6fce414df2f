#!/usr/bin/env python
"""bench.py -- SA-SSD hot path throughput on MI355X (BASELINE.json metric: KITTI-Car inference frames/s).

  python bench.py --gpus N --steps K --warmup W      N > 1 without a torchrun environment: bench.py re-executes itself
                                                     under torch.distributed.run with N ranks (one per GPU, RCCL)

A "step" = one whole pass of the hot path over one batch of synthetic frames, raw points resident in HBM ->
voxelize -> 7 rulebooks -> 14 sparse convs -> densify -> BEVNet -> heads -> anchors mask -> decode/filter -> PSWarp
-> rescore + rotated NMS -> detections in HBM.  The whole frame is ONE hipGraph launch (sassd_graph_*).  Frames shard
across ranks with no data-path collective (weak scaling); value = frames of all ranks / max-over-ranks time.

  --config car    BASELINE configs[1] (HEADLINE, default): car_cfg inference, batch 1, fp32, K21 frames
  --config multi  configs[3]: multi_cfg (Car+Ped+Cyclist) inference, batch 8
  --config waymo  configs[4] shape, inference side: 180k-point frames, 0.1 m voxels, batch 4 per GPU
  --mode train    the training record alone: configs[2] (car_cfg, batch 2 per GPU, bf16, DDP); with --config waymo
                  configs[4] (180k-point frames, 0.1 m voxels, batch 4 per GPU)

The default run (car) measures BOTH halves of BASELINE.json's metric: after the inference timing every rank runs the
configs[2] training step (--train-steps / --train-warmup) and the record -- samples/s, ms per step (max over ranks and
per rank), all-reduce time, the six loss terms, the roofline of the dominant training kernel -- rides in the same JSON
line as `train` (--no-train skips it).

By default three frames are in flight per GPU (--inflight): independent plans / graphs on separate HIP streams, so one
frame's latency-bound sparse / post stages overlap another frame's MFMA-bound BEV stage; each frame is still a batch-1
pass.  `fps_sequential` (one graph after the other on one stream, no host sync) and `latency_ms_sync_per_frame` (host
sync + result read-back per frame) are printed next to it.

The JSON line also carries:
  roofline         dominant kernel (the 36-GEMM launch of a BEV 3x3 layer, Winograd F(4x4,3x3); fp32 products computed as
                   8 bf16 piece products each on the bf16 MFMA over exactly-split operands):
                   `achieved` / `frac` = the flops the kernel EXECUTES on the MFMA pipe (1/4 of the direct convolution,
                   x 8 bf16 multiply-adds per fp32 one) over its mean launch duration (HIP events on the launch stream)
                   vs the 2500 TF bf16-MFMA peak (`frac_of_fp32_mfma_peak`: the same launch against the 157.3 TF
                   fp32-MFMA peak; --wino4-cfg 1 runs the fp32-MFMA kernel of rounds 2-3);
                   `layer.effective` = the direct-convolution flops of SURVEY 8(d) over the whole layer (input
                   transform + GEMM + output transform).
  roofline_sparse  7 rulebooks + 14 sparse convs against the HBM roofline (B_gs bytes of SURVEY 8d), timed as a
                   hipGraph of exactly that segment.
  traffic          PMC counters cannot be collected inside this process: they come from the committed records under
                   profiles/, each stamped with the hash of the kernel sources it was measured on; a record whose stamp
                   differs from the sources of the library in use is dropped (`traffic_measured_at` says so).
  cpu_baseline     the CPU oracle (a faithful port: C voxelizer / NMS + torch-CPU sparse and dense convs) on this box's
                   host cores, 3 warm-ups + >= 20 timed frames, median + per-stage ms (rank 0, N = 1, car only).
"""
import argparse
import json
import os
import sys
import time

# `--hw-queues N` (opt-in): GPU_MAX_HW_QUEUES for the HIP runtime, which multiplexes a process's streams onto that many hardware
# queues (default 4) and reads the variable when it loads, i.e. before `import torch`.  The frame is one hipGraph with two
# branches and `--inflight` of them run at once -- six streams at the default 3.  Measured on one box
# (profiles/r06_late_experiments.txt): 4 queues 807-814 frames/s, 8 queues 839-842, 2 queues 700; multi_cfg 851 -> 894,
# Waymo-scale 397 -> 417.  NOT the default: with 8 queues the sequential rate drops 1.5 %, eager launches with frames in flight
# 791 -> 755, a one-rank DDP step 260 -> 244 samples/s, and the per-stage event timings of the full default line came out doubled
# in two collections for a reason that was not found -- a runtime setting with unexplained side effects stays the caller's choice.
if "--hw-queues" in sys.argv:
    os.environ["GPU_MAX_HW_QUEUES"] = str(int(sys.argv[sys.argv.index("--hw-queues") + 1]))

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import sassd  # noqa: E402
from sassd import synth  # noqa: E402
from sassd.pipeline import InferencePlan  # noqa: E402

PEAK_F32_MFMA_TF = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PROFILE_TAG = "r06"                # profiles/<tag>_*: the PMC records of this round (tools/gpu_profiles_r6.sh, collect_profiles.py)
PEAK_BF16_MFMA_TF = 2500.0        # MI355X_MICROARCH.md: dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16)
SPLIT_SLOTS = 8                   # bf16 multiplies the split GEMM spends per fp32 product (8 of the 9 piece products)
PEAK_HBM_GBS = 8000.0             # HBM3E spec (6.29 TB/s measured float4 copy)
MEASURED_HBM_GBS = 6290.0
PUBLISHED_FPS = 25.0              # /root/reference/readme.md:2 "can run at 25 FPS" 
PYRAMID_DEFAULT = "levels"        # rulebook pyramid form of the plans: "levels" (fill + 2 launches per level) | "persistent"


def build_model(seed=0, dev=None, config="car"):
    """Seeded random-init SA-SSD for the workload + anchors; with `dev` the classification head is rescaled on the HIP
    pipeline so that ~10^2 anchors pass the guided-anchor threshold.  No CPU oracle involved."""
    w = synth.workload(config)
    model, cfg = synth.build_detector_for(w, seed)
    if dev is not None:
        # calibration frame: a sparse KITTI crop (3000 pts) as in the parity tests; a full Waymo-scale frame (the anchor
        # mask of a 30k-point crop would under-count the candidates of a 180k-point frame)
        # calibrated on a frame of the workload's own kind, so that K ~ 10^2 anchors pass the guided-anchor
        # threshold on the frames that are actually measured (candidate counts are printed in the JSON line)
        synth.calibrate_cls_head_on_device(model, w, dev, w["frame"](11), target_count=200 if config != "waymo" else 600)
    return model, w


def cpu_baseline(model, w, warm=3, runs=20, budget_s=60.0, gpu_plan=None):
    """BASELINE.md section 4 protocol on the CPU oracle (imported HERE only: the checker, timed as the baseline).
    gpu_plan: the first frame is also run through the HIP pipeline and the two feature maps are compared -- `parity` in the
    record (the oracle used as the checker it is: voxel rows bit-equal, sparse / BEV feature error relative to the map's
    maximum; the parity TESTS hold the bars, this prints where the line's own build stands against them)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    from oracle import clib, nets as onets
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    torch.set_num_threads(min(torch.get_num_threads(), 16))      # oneDNN / index_add scale poorly past ~16 threads here
    cal, an, bv = w["cal"], w["anchors"], w["anchors_bv"]
    vx, bev, head, ps = H.oracle_params(sd)
    stages = {k: [] for k in ("voxelize", "sparse", "bev_heads", "post")}
    total = []
    t_start = time.time()
    i = 0
    while i < warm + runs and (i < warm + 5 or time.time() - t_start < budget_s):
        pts = synth.k21(100 + i)
        t0 = time.perf_counter()
        v, c, n = clib.points_to_voxel(pts, cal["voxel_size"], cal["pc_range"], cal["max_points"], True, cal["max_voxels"])
        feats = clib.voxel_mean(v, n)
        coors = np.concatenate([np.zeros((len(c), 1), np.int32), c], 1)
        t1 = time.perf_counter()
        x3, idx3, shape3, *_ = onets.vxnet_forward(feats, coors, cal["sparse_shape"], 1, vx)
        t2 = time.perf_counter()
        x, conv6 = onets.bevnet_forward(onets.densify(x3, idx3, shape3, 1), bev)
        box, cls, dirp = onets.ssd_head_forward(x, head, 1)
        t3 = time.perf_counter()
        if i == 0 and gpu_plan is not None:
            dev = gpu_plan.dev
            gpu_plan.run_from_points([torch.from_numpy(pts).to(dev)])
            torch.cuda.synchronize()
            n0, n3 = int(gpu_plan.n[0].item()), int(gpu_plan.n[3].item())
            parity = dict(frame="synth.k21(100)", voxel_rows_bit_equal=bool(np.array_equal(gpu_plan.idx[0][:n0].cpu().numpy(), coors)),
                          level3_rows_bit_equal=bool(np.array_equal(gpu_plan.idx[3][:n3].cpu().numpy(), idx3)),
                          sparse_feature_err_over_max=float((gpu_plan.sp_out[:n3].cpu() - x3).abs().max() / x3.abs().max()),
                          bev_feature_err_over_max=float((gpu_plan.x.cpu() - x).abs().max() / x.abs().max()),
                          bars="sparse 1e-5, BEV 2e-5 of the maximum (tests/test_gpu_pipeline.py)")
        mask = onets.anchors_mask(c, bv, cal["voxel_size"], cal["pc_range"], cal["grid_xyz"], 1)[None]
        guided = onets.guided_anchors(box, cls, dirp, torch.from_numpy(an).view(1, -1, 7), torch.from_numpy(mask), 1, 0.1)
        logits, _ = onets.pswarp_forward(conv6, ps, [g[0] for g in guided])
        [onets.rescore(g[0], lg, g[1], 0.3, 0.1) for g, lg in zip(guided, logits)]
        t4 = time.perf_counter()
        if i >= warm:
            for k, dt in zip(("voxelize", "sparse", "bev_heads", "post"), (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
                stages[k].append(dt * 1e3)
            total.append((t4 - t0) * 1e3)
        i += 1
    med = float(np.median(total))
    extra = {} if gpu_plan is None else dict(parity=parity)
    return dict(extra, value=round(1e3 / med, 4), unit="frames/s", cores=torch.get_num_threads(), kind="port",
                ms_per_frame_median=round(med, 1), ms_p10=round(float(np.percentile(total, 10)), 1),
                ms_p90=round(float(np.percentile(total, 90)), 1),
                stage_ms_median={k: round(float(np.median(v)), 2) for k, v in stages.items()},
                sample="%d warm-ups + %d timed K21 frames (21.5k pts), whole path on the CPU oracle: C voxelizer + "
                       "rotated NMS on 1 thread, torch-CPU gather/mm/index_add sparse convs and oneDNN conv2d on %d "
                       "threads; value = 1000 / median ms" % (warm, len(total), torch.get_num_threads()))


def synth_gt_on_points(cloud, seed, n=8, config="car"):
    """Car-sized ground-truth boxes centred on occupied voxels of the frame (so anchors get positive matches and the
    localisation / direction losses are exercised): x,y,z bottom centre, w,l,h,ry."""
    r = np.random.default_rng(seed)
    c = cloud[r.choice(len(cloud), n, replace=False), :3]
    b = np.zeros((n, 7), np.float32)
    if config == "waymo":
        b[:, 0], b[:, 1] = np.clip(c[:, 0], -70, 70), np.clip(c[:, 1], -70, 70)
    else:
        b[:, 0], b[:, 1] = np.clip(c[:, 0], 3, 67), np.clip(c[:, 1], -37, 37)
    b[:, 2] = r.uniform(-1.9, -1.5, n)
    b[:, 3], b[:, 4], b[:, 5] = r.uniform(1.5, 1.8, n), r.uniform(3.5, 4.4, n), r.uniform(1.4, 1.7, n)
    b[:, 6] = r.choice([0.0, 1.57, -1.57, 3.1], n) + r.uniform(-0.2, 0.2, n)
    return b


def csrc_hash():
    from sassd import _C
    return _C.csrc_hash()


def stamped_traffic(fname):
    """(record, measured_at) of a committed PMC traffic file; the record is dropped (None) when the file carries no
    `csrc_hash` or one that differs from the sources of the library in use -- a stale counter must not ride along."""
    path = os.path.join(ROOT, "profiles", fname)
    if not os.path.exists(path):
        return None, None
    rec = json.load(open(path))
    at = rec.get("csrc_hash")
    if at is None or at != csrc_hash():
        return None, "%s: measured at csrc %s, library built from %s -> dropped" % (fname, at, csrc_hash())
    return rec, "%s @ csrc %s" % (fname, at)


def train_measure(args, dev, rank, world, config="car", precision="bf16", steps=40, warmup=8, batch=0, frames=16,
                  fused_bn=None):
    """_train_measure with the process-wide training switches (BEV precision, fused sparse BatchNorm) restored afterwards,
    whatever happens inside: the inference half of the default line runs in the same process."""
    from sassd import autograd as AG, spconv as SP
    prev = (AG.bev_precision(), SP.SparseSequential.fuse_bn_relu)
    try:
        return _train_measure(args, dev, rank, world, config, precision, steps, warmup, batch, frames, fused_bn)
    finally:
        AG.set_bev_precision(prev[0])
        SP.SparseSequential.fuse_bn_relu = prev[1]


def _train_measure(args, dev, rank, world, config, precision, steps, warmup, batch, frames, fused_bn):
    """The training half of the BASELINE metric.  config "car": BASELINE configs[2] (car_cfg training, batch 2 / GPU,
    bf16 MFMA operands in the BEV convs); "waymo": configs[4] (180k-point frames, 0.1 m voxels, batch 4 / GPU).
    A step = device voxelize + anchor masks + rulebooks (side stream) + forward_train + backward + bucketed gradient
    all-reduce (RCCL) + fused clip / AdamW update + one-launch weight re-pack.  Returns the `train` record
    (rank 0; None on the other ranks).  Reference loop: tools/train_utils/__init__.py:36-61, DDP wrap tools/train.py:78."""
    from sassd import dist as D, train, autograd as AG
    AG.set_bev_precision(precision)
    if fused_bn is not None:
        from sassd import spconv as SP
        SP.SparseSequential.fuse_bn_relu = bool(fused_bn)
    w = synth.workload(config)
    model, cfg = synth.build_detector_for(w, 0, train=True, cls_bias=-3.0)
    model = model.to(dev)
    B = batch if batch > 0 else (4 if config == "waymo" else 2)
    anchors = dict(Car=torch.from_numpy(w["anchors"]).to(dev))
    anchors_bv = dict(Car=torch.from_numpy(w["anchors_bv"]).to(dev))
    opt = train.build_optimizer(model, cfg.optimizer, world)
    sched = train.build_scheduler(opt, 5 * steps + warmup, 1, cfg.optimizer, cfg.lr_config)   # (up to five trials)
    sync = train.GradSync(opt.flat, time_comm=True, force=getattr(args, "force_ddp", False))
    nf = max(min(frames, 8) if config == "waymo" else frames, B)
    host = [w["frame"](rank * 1000 + i) for i in range(nf)]
    clouds = [torch.from_numpy(p).to(dev) for p in host]
    ngt = 12 if config == "waymo" else 8
    gts = [torch.from_numpy(synth_gt_on_points(p, rank * 1000 + i, ngt, config)).to(dev) for i, p in enumerate(host)]
    types = [np.array(["Car"] * ngt) for _ in range(nf)]
    cal = w["cal"]

    def make_batch(i):
        ids = [(i * B + j) % nf for j in range(B)]
        return train.device_batch([clouds[k] for k in ids], [gts[k] for k in ids], [types[k] for k in ids], ["Car"],
                                  anchors, anchors_bv, cal["voxel_size"], cal["pc_range"], max_points=cal["max_points"],
                                  max_voxels=cal["max_voxels"], model=model)

    make_next = train.SideStreamPrefetch(make_batch)        # data preparation on its own stream: its host reads of
    state = {"batch": make_next(0)}                         # row counts do not drain the training step

    def one(i):
        # the next batch (device voxelize, anchor masks, rulebooks -- the host syncs) is built between this step's
        # forward and backward
        loss, terms, state["batch"] = train.train_one_iter(model, opt, sched, sync, state["batch"], i,
                                                           prefetch=lambda: make_next(i + 1))
        return loss, terms

    def barrier():
        torch.cuda.synchronize()
        D.barrier()
        torch.cuda.synchronize()

    for i in range(warmup):
        loss, _ = one(i)
    comm = []
    it0 = [warmup]

    def trial():                                        # EXACTLY `steps` steps between barrier + synchronize
        barrier()
        t0 = time.perf_counter()
        last = None
        for i in range(steps):
            last = one(it0[0] + i)
            if sync.on and i >= steps - 5:
                comm.append(sync.comm_ms())             # (synchronises on the exchange's end event: last steps only)
        torch.cuda.synchronize()
        dl = time.perf_counter() - t0
        barrier()
        it0[0] += steps
        return D.allreduce_max(time.perf_counter() - t0, dev), dl, last

    trials = [trial()]
    if trials[0][0] < 0.5:                              # short run (40 steps = 0.35 s): five trials, the median reported
        trials += [trial() for _ in range(4)]
    trials_dt = sorted(t[0] for t in trials)
    dt = trials_dt[len(trials_dt) // 2]
    dt_local = sorted(t[1] for t in trials)[len(trials) // 2]
    loss, terms = trials[-1][2]
    per_rank = D.allgather_float(dt_local / steps * 1e3, dev) if world > 1 else [dt_local / steps * 1e3]
    voxels = int(sum(v.shape[0] for v in state["batch"]["voxels"]))
    # dominant kernel of the training step, timed live with events on the launch stream: the 3x3 BEV conv (14 launches
    # per step, forward + data gradient) -- bf16 direct kernel where the shape allows, else the fp32 Winograd F(4x4) layer
    from sassd import kernels as K
    H, W = model.neck.sparse_shape[1] // 8, model.neck.sparse_shape[2] // 8
    xb = torch.randn(B, 256, H, W, device=dev)
    wb = torch.randn(256, 256, 3, 3, device=dev) / 48
    flops = 2.0 * B * H * W * 256 * 256 * 9
    bf16_conv = precision == "bf16" and K.conv2d_bf16_supported(256, 256, H, W)
    if bf16_conv:
        pk = K.conv2d_bf16_pack_weight(wb)
        run, kname, peak = (lambda: K.conv2d_bf16_fwd(xb, pk, 256)), "conv2d_bf16_kernel (v_mfma_f32_32x32x16_bf16)", 2500.0
    else:
        pk = K.conv2d_wino4_pack_weight(wb)
        run = lambda: K.conv2d_wino4_fwd(xb, pk, 256, None, None)       # noqa: E731
        kname, peak = "wino4_in + wino4_gemm + wino4_out (fp32 MFMA, Winograd F(4x4): executed flops = direct / 4)", 157.3
        flops /= 4.0
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    kms = e0.elapsed_time(e1) / 20
    del xb, wb
    if rank != 0:
        return None
    sps = steps * B * world / dt
    traffic, at = (None, None)
    if bf16_conv and B == 2 and config == "car":
        rec, at = stamped_traffic(PROFILE_TAG + "_bf16_conv_hbm_traffic.json")
        traffic = rec["traffic_bytes_per_launch"] if rec else None
    roof = dict(bound="mfma", kernel=kname, achieved=round(flops / kms / 1e9, 1), peak=peak, unit="TFLOP/s",
                frac=round(flops / kms / 1e9 / peak, 4), ms_per_launch=round(kms, 4), traffic=traffic,
                traffic_measured_at=at,
                note="flops executed by one 256->256 3x3 BEV layer at batch %d (%dx%d map) / mean layer time (HIP events, 20 "
                     "launches on the launch stream); 14 such layers per step (forward + data gradient)" % (B, H, W))
    bev = ("bf16 MFMA operands in the dense convs (BEV 3x3 + 1x1 and the head convs: fwd/dgrad/wgrad; fp32 accumulation, master weights and activations), "
           "fp32 sparse trunk" if bf16_conv else
           "bf16 weight gradients, fp32 Winograd forward / data gradient (the bf16 direct conv needs W %% 16 == 0, the "
           "BEV map is %dx%d), fp32 sparse trunk" % (H, W) if precision == "bf16" else "fp32")
    desc = ("configs/car_cfg.py training, batch=%d/GPU, %s, synthetic lidar64 K21 frames + 8 synthetic car boxes/frame "
            "on occupied voxels, adam_onecycle, grad clip 10" % (B, bev)) if config == "car" else (
            "Waymo-scale synthetic training (BASELINE configs[4]): car head, batch=%d/GPU, 180000 pts/frame, 0.1x0.1x0.15 "
            "m voxels (grid 40x1504x1504, %d active voxels in the last batch), BEV 188x188, %s, 12 synthetic car "
            "boxes/frame on occupied voxels, adam_onecycle, grad clip 10" % (B, voxels, bev))
    return {
        "metric": "%s training samples/sec (whole job)" % ("KITTI-Car" if config == "car" else "Waymo-scale synthetic"),
        "value": round(sps, 3), "unit": "samples/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 3),
        "trials": {"n": len(trials_dt), "steps_each": steps, "reported": "median",
                   "ms_per_step_min": round(trials_dt[0] / steps * 1e3, 3), "ms_per_step_max": round(trials_dt[-1] / steps * 1e3, 3)},
        "ms_per_step_per_rank": [round(v, 3) for v in per_rank],
        "allreduce_ms": None if not comm else round(float(np.mean([c for c in comm if c is not None])), 3),
        "allreduce_note": "first gradient-bucket launch -> last bucket complete on the compute stream (4 buckets launched "
                          "from backward hooks, so most of it overlaps the sparse backward); null at 1 GPU",
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16" if precision == "bf16" else "f32", "data": "synthetic",
        "config": {"workload": desc, "global_batch": B * world,
                   "parallelism": "ddp x%d (flat-gradient RCCL all-reduce in 4 buckets / step%s)" %
                                  (world, "; one-rank communicator forced" if (sync.on and world == 1) else "")},
        "roofline": roof,
        "final_loss": round(float(loss), 4), "loss_terms": {k: round(float(v), 4) for k, v in terms.items()},
        "final_loss_note": "after %d steps of a one-cycle schedule laid out for %d (warm-up + up to five trials): a sanity value "
                           "(finite, six non-zero terms), not comparable across --steps or with rounds 1-3, whose schedule "
                           "was sized for one trial" % (warmup + len(trials_dt) * steps, 5 * steps + warmup)}


def main_train(args):
    """--mode train: the training record as the JSON line (configs[2] by default, configs[4] with --config waymo)."""
    from sassd import dist as D
    torch.cuda.set_device(D.env_world()[1])
    rank, local_rank, world = D.init("nccl", force_single=args.force_ddp)
    dev = torch.device("cuda", local_rank)
    if args.spconv_cfg:
        from sassd import kernels as K0
        K0.DEFAULT_CFG["spconv"] = K0.spconv_cfg(args.spconv_cfg)     # host-side default of the binding, passed per call
    out = train_measure(args, dev, rank, world, "waymo" if args.config == "waymo" else "car", args.precision,
                        args.steps, args.warmup, args.batch if args.batch > 1 else 0, args.frames,
                        False if args.torch_bn else None)
    if rank == 0:
        print(json.dumps(out))


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` with no torchrun environment: become N ranks (one per GPU) of one node."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    os.execvpe(cmd[0], cmd, env)


def timed_graph_ms(plan, batch, reps=30):
    """Mean replay time of plan.graph on the current stream (events around `reps` back-to-back replays)."""
    plan.stage_inputs(batch)
    for _ in range(3):
        plan.graph.launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        plan.graph.launch()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def infer_extra(dev, config, steps=60, warmup=10, inflight=3, pyramid_persistent=False):
    """configs[3] / configs[4] (inference side) as a compact record of the DEFAULT line (`infer_multi`, `infer_waymo`), so that
    the driver's run measures them too: the same measurement as the headline -- `steps` hipGraph replays of the whole frame
    batch, `inflight` plans on separate streams, bracketed by synchronize, median of three trials -- plus the sparse segment
    (rulebooks + 14 sparse convs as its own hipGraph) against the HBM roofline."""
    model, w = build_model(0, dev, config)
    B, S = w["batch"], max(1, inflight)
    sd = model.state_dict()
    def mk(overlap):
        return InferencePlan(sd, batch_size=B, anchors=w["anchors"], anchors_bv=w["anchors_bv"], device=dev,
                             pyramid_persistent=pyramid_persistent, overlap=overlap, **w["plan"])
    plans = [mk(S == 1) for _ in range(S)]              # frames in flight: one-branch graphs (see main())
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    clouds = [torch.from_numpy(w["frame"](i)).to(dev) for i in range(max(8, B))]

    def batch_of(i):
        return [clouds[(i * B + j) % len(clouds)] for j in range(B)]
    for pl, st in zip(plans, streams):
        with torch.cuda.stream(st):
            pl.capture(w["points_cap"])
    torch.cuda.synchronize()

    def step(i):
        with torch.cuda.stream(streams[i % S]):
            plans[i % S].run_graph(batch_of(i))
    for i in range(max(warmup, S)):
        step(i)
    torch.cuda.synchronize()
    for pl in plans:
        assert int(pl.status.item()) == 0, "pipeline status 0x%x" % int(pl.status.item())
    dts = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        torch.cuda.synchronize()
        dts.append(time.perf_counter() - t0)
    dt = float(np.median(dts))
    plan = plans[0]
    ndet, ncand = int(plan.det["counts"].sum().item()), int(plan.df["counts"].sum().item())
    for pl in plans:
        assert int(pl.status.item()) == 0, "pipeline status 0x%x after the timed runs" % int(pl.status.item())
    del plans
    torch.cuda.empty_cache()
    plans = plan = mk(True)                             # the sparse segment as its own graph: the two-branch form, as in main()
    with torch.cuda.stream(streams[0]):
        plan.run_from_points(batch_of(0))
        torch.cuda.synchronize()
        work = plan.sparse_work()
        plan.capture(w["points_cap"], stages=("sparse",))
        plan.stage_inputs(batch_of(0))
        sp_ms = timed_graph_ms(plan, None, reps=20)
    sp_gbs = (work["bytes_gs"] + work["rulebook_bytes"]) / (sp_ms * 1e-3) / 1e9
    del plans, plan
    torch.cuda.empty_cache()
    return {"metric": "%s inference frames/sec" % config, "value": round(steps * B / dt, 3), "unit": "frames/s",
            "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps, "warmup": warmup, "trials": 3,
            "config": {"workload": "%s, batch=%d, random-init SA-SSD weights, points resident in HBM" % (w["desc"], B),
                       "frames_per_step_per_gpu": B, "frames_in_flight": S},
            "roofline_sparse": {"bound": "hbm", "ms": round(sp_ms, 4), "achieved": round(sp_gbs, 1), "unit": "GB/s",
                                "frac": round(sp_gbs / PEAK_HBM_GBS, 4),
                                "frac_of_measured_copy_peak": round(sp_gbs / MEASURED_HBM_GBS, 4),
                                "bytes_gs": work["bytes_gs"], "bytes_min": work["bytes_min"],
                                "rulebook_bytes": work["rulebook_bytes"], "rows": work["n"]},
            "detections_last_frame": ndet, "guided_anchor_candidates_last_frame": ncand}


def child_record(argv, timeout_s):
    """One more record of the default line, measured by THIS script in a child process (`python bench.py <argv>`) and read back
    from its JSON line.  World size 1 only.  Why a child: the same measurement inside the process that has just captured and
    replayed the frame graphs ran slower for reasons outside the kernels -- `train` 238-251 instead of 276 samples/s, Waymo-scale
    inference 383 instead of 411 frames/s on the same box (how the runtime places a long-lived process's streams on its hardware
    queues) -- and a record of the line should be what `python bench.py <argv>` prints when run alone."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__)] + list(argv)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout_s, env=env)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if p.returncode != 0 or not lines:
        raise RuntimeError("child %s exited %d: %s" % (" ".join(argv), p.returncode, p.stderr.strip().splitlines()[-1:] or ""))
    rec = json.loads(lines[-1])
    rec["measured_by"] = "child process: python bench.py " + " ".join(argv)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=("infer", "train"), default="infer",
                    help="infer = BASELINE configs[1] (headline); train = configs[2] shape, extra measurement")
    ap.add_argument("--config", choices=("car", "multi", "waymo"), default="car",
                    help="car = configs[1] (headline); multi = configs[3] (batch 8); waymo = configs[4] shape (batch 4)")
    ap.add_argument("--precision", choices=("bf16", "fp32"), default="bf16",
                    help="--mode train: arithmetic of the dense BEV convolutions (BASELINE configs[2] trains in bf16)")
    ap.add_argument("--torch-bn", action="store_true", help="--mode train: torch's BatchNorm1d + ReLU for the sparse blocks "
                    "instead of the fused kernels (sassd.spconv.SparseSequential.fuse_bn_relu = False; A/B)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--force-ddp", action="store_true", help="N = 1: initialise a ONE-RANK nccl (RCCL) communicator and run the "
                    "training step's bucketed gradient all-reduce through it (hook-launched async collectives on the HIP "
                    "stream), so that `train.allreduce_ms` is measured on a single-GPU box")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="default (car) run: skip the `train` record (BASELINE "
                    "configs[2], the training half of the metric) that follows the inference measurement")
    ap.add_argument("--train-steps", type=int, default=40, help="timed training steps of the default run's `train` record")
    ap.add_argument("--train-warmup", type=int, default=8)
    ap.add_argument("--frames", type=int, default=16, help="distinct synthetic frames cycled through")
    ap.add_argument("--inflight", type=int, default=3, help="frames in flight: independent plans on separate HIP "
                    "streams, so one frame's latency-bound sparse stage overlaps another frame's MFMA-bound BEV stage")
    ap.add_argument("--batch", type=int, default=0, help="frames per step per GPU (default: the config's own -- car 1, "
                    "multi 8, waymo 4)")
    ap.add_argument("--spconv-cfg", type=int, default=0, help="workgroup geometry of the sparse gather-GEMM-scatter kernel "
                    "(A/B): 0 = default (8 waves, 128 KB of LDS slabs: one workgroup per CU), 1 = 4 waves / 64 KB (two per "
                    "CU, and room beside a BEV GEMM workgroup of another frame in flight)")
    ap.add_argument("--wino4-cfg", type=int, default=0, help="geometry of the Winograd GEMM (A/B): 0 = fp32 products on the "
                    "bf16 MFMA over split operands (default), 1 = the fp32 MFMA (the default of rounds 2-3)")
    ap.add_argument("--wino4-dbg", type=int, default=0, help="ablation flags of the Winograd cfg word for the whole frame (A/B; "
                    "256 = the fused output->input transform on 256 threads, the rounds 3-5 form)")
    ap.add_argument("--eager", action="store_true", help="issue the ~80 launches per frame from the host instead of "
                    "replaying the captured hipGraph (A/B)")
    ap.add_argument("--pyramid", choices=("levels", "persistent"), default=PYRAMID_DEFAULT, help="rulebook pyramid: two "
                    "launches per level, or ONE persistent launch with in-launch grid barriers (A/B)")
    ap.add_argument("--rb-sync", default="0,1,2,3", help="rulebook levels at which the feature stream joins the coordinate stream "
                    "(0,1,2,3 = one wait per level, the default; A/B: 1,3 or 3)")
    ap.add_argument("--dense-conv0", action="store_true", help="BEV conv0 on every tile instead of the active tiles of the "
                    "sparse map only (A/B)")
    ap.add_argument("--hw-queues", type=int, default=0, help="GPU_MAX_HW_QUEUES for the HIP runtime (read before torch loads: see "
                    "the top of this file); 0 = leave the environment alone")
    ap.add_argument("--side-stream", action="store_true", help="the plans in flight keep their coordinate side stream: every frame "
                    "graph has two branches (the form of rounds 2-6; A/B).  Default: ONE branch per frame in flight -- "
                    "rulebooks / anchor masks in front of the feature path on the frame's own stream")
    ap.add_argument("--no-extra", action="store_true", help="default (car) run: skip the `infer_multi` / `infer_waymo` / "
                    "`train_waymo` records (BASELINE configs[3] / [4]) that follow the headline measurement")
    args = ap.parse_args()
    env_world = int(os.environ.get("WORLD_SIZE", "0") or 0)
    if args.gpus > 1 and env_world == 0:
        respawn_under_torchrun(args)                     # does not return
    if env_world and env_world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, env_world))
    if args.mode == "train":
        return main_train(args)

    from sassd import dist as D
    torch.cuda.set_device(D.env_world()[1])
    rank, local_rank, world = D.init("nccl", force_single=args.force_ddp)     # RCCL over xGMI; only the barrier + max-time
    torch.cuda.set_device(local_rank)                                         # reduction (and the DDP exchange) use it
    dev = torch.device("cuda", local_rank)

    from sassd import kernels as K0
    plan_cfg = dict(spconv_cfg=K0.spconv_cfg(args.spconv_cfg), wino4_cfg=K0.wino4_cfg(args.wino4_cfg, args.wino4_dbg),   # per-call words
                    rb_sync_levels=tuple(int(v) for v in args.rb_sync.split(",")), skip_inactive_tiles=not args.dense_conv0)
    model, w = build_model(0, dev, args.config)
    B = args.batch if args.batch > 0 else w["batch"]
    S = max(1, args.inflight)
    sd = model.state_dict()

    # Frames in flight are ONE-BRANCH graphs (round 6, last day).  A plan with the coordinate side stream captures a graph with two
    # branches; three of them in flight are six streams on the runtime's four hardware queues, and graphs that share a queue run
    # their kernels in each other's order: 792-808 frames/s.  Without the side stream a frame is one linear graph on its own queue
    # and the overlap of latency-bound and MFMA-bound stages comes from the OTHER frames: 895-942 frames/s on the same boxes (and
    # 879-924 for every placement of the three streams in torch's pool: the result does not hang on the stream -> queue mapping);
    # one frame alone is 2 % slower that way (664 against 677 sequential: its rulebooks no longer overlap its own first convs),
    # the host-synced latency is the same 1.59 ms.  `--side-stream` restores the two-branch form (profiles/r06_late_experiments.txt).
    one_branch = S > 1 and not args.side_stream and not args.eager

    def new_plan(overlap=True):
        return InferencePlan(sd, batch_size=B, anchors=w["anchors"], anchors_bv=w["anchors_bv"], device=dev,
                             pyramid_persistent=args.pyramid == "persistent", overlap=overlap, **plan_cfg, **w["plan"])

    plans = [new_plan(overlap=not one_branch) for _ in range(S)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    plan = plans[0]
    nfr = max(args.frames if args.config != "waymo" else min(args.frames, 8), B)
    clouds = [torch.from_numpy(w["frame"](rank * 1000 + i)).to(dev) for i in range(nfr)]

    def batch_of(i):
        return [clouds[(i * B + j) % len(clouds)] for j in range(B)]

    def barrier():
        torch.cuda.synchronize()
        D.barrier()
        torch.cuda.synchronize()

    if not args.eager:
        for pl, st in zip(plans, streams):
            with torch.cuda.stream(st):
                pl.capture(w["points_cap"])
        torch.cuda.synchronize()

    def step(i):
        with torch.cuda.stream(streams[i % S]):
            if args.eager:
                plans[i % S].run_from_points(batch_of(i))
            else:
                plans[i % S].run_graph(batch_of(i))

    for i in range(max(args.warmup, S)):
        step(i)
    torch.cuda.synchronize()
    for pl in plans:
        st = int(pl.status.item())
        assert st == 0, "pipeline status 0x%x" % st

    # EXACTLY K steps between barrier + synchronize on both sides, max over ranks.  A trial shorter than half a second
    # (the driver's 20 steps are 30 ms) is repeated -- five trials at least, each of them K steps bracketed the same
    # way -- and the MEDIAN trial is reported; `steps` / `ms_per_step` keep their meaning, `trials` lists the spread.
    def trial():
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        barrier()
        return D.allreduce_max(time.perf_counter() - t0, dev)

    trial_dts = [trial()]
    if trial_dts[0] < 0.5:
        ntr = int(D.allreduce_max(float(max(5, min(25, int(np.ceil(1.5 / max(trial_dts[0], 1e-4)))))), dev))
        trial_dts += [trial() for _ in range(ntr - 1)]
    dt = float(np.median(trial_dts))
    for pi, pl in enumerate(plans):
        st = int(pl.status.item())
        assert st == 0, "pipeline status 0x%x on plan %d after the timed trials" % (st, pi)
    ndet = int(plan.det["counts"].sum().item())
    ncand = int(plan.df["counts"].sum().item())
    fps = args.steps * B * world / dt

    # ---- everything below is measurement detail on top of the timed region -----------------------------------
    # sequential: one graph after the other on ONE stream, no host sync in between
    # ONE frame at a time is the other deployment: there the frame's own coordinate work should overlap its first convolutions, i.e.
    # the two-branch graph (2 % faster sequentially than the one-branch graph the frames in flight use) -- these two numbers and
    # everything below are taken on such a plan
    seq_steps = max(20, min(args.steps, 100))
    lat_plan = plan
    if one_branch:
        lat_plan = new_plan()
        with torch.cuda.stream(streams[0]):
            lat_plan.capture(w["points_cap"])
        torch.cuda.synchronize()
    with torch.cuda.stream(streams[0]):
        for i in range(3):
            lat_plan.run_from_points(batch_of(i)) if args.eager else lat_plan.run_graph(batch_of(i))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(seq_steps):
            lat_plan.run_from_points(batch_of(i)) if args.eager else lat_plan.run_graph(batch_of(i))
        torch.cuda.synchronize()
        seq_ms = (time.perf_counter() - t1) / seq_steps * 1e3
        # latency mode: host sync + result read-back per frame
        nlat = max(10, min(50, args.steps))
        t1 = time.perf_counter()
        for i in range(nlat):
            lat_plan.run_from_points(batch_of(i)) if args.eager else lat_plan.run_graph(batch_of(i))
            lat_plan.results()
        lat_ms = (time.perf_counter() - t1) / nlat * 1e3

    # isolated eager pass with HIP events on the launch stream: per-stage / per-kernel durations of one frame at a time
    iso_plan = lat_plan if one_branch else new_plan()
    iso_plan.prof = {}
    with torch.cuda.stream(streams[0]):
        for i in range(30):
            iso_plan.run_from_points(batch_of(i))
        torch.cuda.synchronize()
    prof_iso, iso_plan.prof = iso_plan.prof, None
    work = iso_plan.sparse_work()                              # of the last frame processed

    # the sparse segment (7 rulebooks + 14 sparse convs) as its own hipGraph
    sp_ms = sp_convs_ms = sp_pyr_ms = None
    if not args.eager:
        with torch.cuda.stream(streams[0]):
            iso_plan.capture(w["points_cap"], stages=("sparse",))
            iso_plan.stage_inputs(batch_of(29))
            sp_ms = timed_graph_ms(iso_plan, None)
            # the two halves of the segment on their own (the floor model of DESIGN section 8: the convs on the rulebooks the
            # previous replay left behind; the pyramid alone)
            iso_plan.capture(w["points_cap"], stages=("sparse_convs",))
            sp_convs_ms = timed_graph_ms(iso_plan, None)
            iso_plan.capture(w["points_cap"], stages=("pyramid",))
            sp_pyr_ms = timed_graph_ms(iso_plan, None)
            iso_plan.capture(w["points_cap"], stages=("voxelize", "backbone", "tail"))
            frame_ms = timed_graph_ms(iso_plan, batch_of(29))
    # the launches of an F(4x4) layer timed separately on the LIVE buffers the last frame left behind (real activations:
    # zero-filled or dense-random operands clock the chip differently): the GEMM on the transformed input of conv6, the fused
    # output -> input transform and the output transform on its products, the input transform on the densified map
    w4_parts = {}
    if iso_plan.bev[1][5] == 4:
        from sassd import kernels as K
        wp, cout, ks, scale, shift, _ = iso_plan.bev[1]
        wp0 = iso_plan.bev[0][0]
        H_, W_, cm, ws = iso_plan.H, iso_plan.W, iso_plan.cmax, iso_plan.wino4_ws
        chained = any(iso_plan.chain)

        def timed_part(flags, call):
            word = K.wino4_cfg(args.wino4_cfg, flags)
            for _ in range(3):
                call(word)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                call(word)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / 20
        with torch.cuda.stream(streams[0]):
            prev = (scale, shift, True)
            # debug bits: 16 skip input transform, 32 skip GEMM, 64 skip output transform, 128 skip fused transform
            w4_parts["gemm"] = timed_part(128 | 16, lambda c: K.conv2d_wino4_chain(
                None, prev, wp, 256, cout, cm, B, H_, W_, scale, shift, True, None, ws, cfg=c))
            if chained:
                w4_parts["outin_fused"] = timed_part(32, lambda c: K.conv2d_wino4_chain(
                    None, prev, wp, 256, cout, cm, B, H_, W_, scale, shift, True, None, ws, cfg=c))
            w4_parts["out"] = timed_part(128 | 32, lambda c: K.conv2d_wino4_chain(
                None, prev, wp, 256, cout, cm, B, H_, W_, scale, shift, True, iso_plan.act[1], ws, cfg=c))
            w4_parts["in"] = timed_part(32, lambda c: K.conv2d_wino4_chain(
                iso_plan.dense, None, wp0, iso_plan.bev_cin[0], cout, cm, B, H_, W_, scale, shift, True, None, ws, cfg=c))
    headline = args.config == "car" and B == 1
    # ---- the training half of BASELINE.json's metric (configs[2]: car_cfg, batch 2 / GPU, bf16, DDP): every rank takes
    # part (gradient all-reduce over RCCL at N > 1); the record rides in the same JSON line as `train` ------------------
    emit = {}
    if headline and not args.no_train and not args.eager:
        import threading

        def give_up():                                       # a hung collective must not cost the inference line
            if rank == 0 and "line" in emit:
                emit["line"]["train"] = {"error": "training measurement exceeded %d s (hung collective?)" % 420}
                print(json.dumps(emit["line"]), flush=True)
            os._exit(0 if rank == 0 else 1)
        watchdog = threading.Timer(420.0, give_up)
        watchdog.daemon = True
    else:
        watchdog = None
    if rank != 0:
        if watchdog is not None:
            watchdog.start()
            try:
                train_measure(args, dev, rank, world, "car", "bf16", args.train_steps, args.train_warmup)
            finally:
                watchdog.cancel()
        return
    iso_ms = {k: float(np.mean([a.elapsed_time(b) for a, b in v][5:])) for k, v in prof_iso.items()}
    if sp_ms is None:
        sp_ms, frame_ms = iso_ms["sparse"], None
    H, W = plan.H, plan.W
    kind = plan.bev[1][5]                                      # 4 = Winograd F(4x4,3x3), 2 = fused F(2x2), 0 = direct
    conv_iso = float(np.mean([iso_ms["bev_conv%d" % i] for i in range(1, 7)]))      # six identical 256->256 3x3 layers
    conv_flops = 2.0 * 256 * 256 * 9 * H * W * B               # direct-convolution flops of the layer
    exec_flops = conv_flops * {4: 36.0 / 144.0, 2: 16.0 / 36.0, 0: 1.0}[kind]       # what the MFMA pipe executes
    kernel_ms = w4_parts.get("gemm", conv_iso)                 # the dominant KERNEL: the F(4x4) GEMM launch alone
    exec_tf = exec_flops / (kernel_ms * 1e-3) / 1e12
    eff_tf = conv_flops / (conv_iso * 1e-3) / 1e12
    split = kind == 4 and args.wino4_cfg in (0, 11, 12, 13, 14)
    kname = {4: "wino4_gemm_kernel (36 GEMMs 256 x 256 x tiles of the BEV 256->256 3x3 layer, Winograd F(4x4,3x3), " +
                ("fp32 products as 8 bf16 piece products each on v_mfma_f32_32x32x16_bf16, fp32 accumulate)" if split else
                 "fp32 MFMA 32x32x2)"), 2: "conv2d_wino_kernel (BEV 256->256 3x3, fused Winograd F(2x2,3x3), fp32 MFMA 32x32x2)",
             0: "conv2d_kernel (BEV 256->256 3x3, direct, fp32 MFMA 32x32x2)"}[kind]
    # split geometry: the pipe executes SPLIT_SLOTS bf16 multiply-adds per fp32 one, priced against the bf16 MFMA peak
    mfma_mult, mfma_peak = (SPLIT_SLOTS, PEAK_BF16_MFMA_TF) if split else (1, PEAK_F32_MFMA_TF)
    bev_total_ms = sum(iso_ms["bev_conv%d" % i] for i in range(8))
    sp_gbs = (work["bytes_gs"] + work["rulebook_bytes"]) / (sp_ms * 1e-3) / 1e9
    # PMC passes cannot run inside this process: the committed measurement is read -- and dropped unless it was taken on
    # the kernel sources this library was built from (`traffic_measured_at` says which)
    traffic = traffic_at = sp_traffic = sp_traffic_at = None
    if B == 1 and args.config == "car":
        rec, traffic_at = stamped_traffic(PROFILE_TAG + "_wino4_gemm_hbm_traffic.json")
        traffic = rec["traffic_bytes_per_launch"] if rec else None
    if B == w["batch"]:                  # fabric-side bytes of one sparse pass: 2 x FETCH + WRITE
        t, sp_traffic_at = stamped_traffic(PROFILE_TAG + "_sparse_%s_hbm_traffic.json" % args.config)
        if t:
            sp_traffic = int((2 * t["FETCH_SIZE_kb_per_pass_raw"] + t["WRITE_SIZE_kb_per_pass_raw"]) * 1024)
    # issue / wait / MFMA-busy fractions of the sparse-conv kernels from the stamped SQ-counter passes (profiles/)
    sp_counters, sp_counters_at = None, None
    st, sp_counters_at = stamped_traffic(PROFILE_TAG + "_stall_breakdown.json")
    if st:
        sp_counters = {k: dict(v["fraction_of_wave_cycles"], mfma_busy_fraction=v.get("mfma_busy_fraction"))
                       for k, v in st["kernels"].items() if k.startswith("spconv") and k.endswith("@" + args.config)} or None
    out = {
        "metric": "KITTI-Car inference frames/sec (whole job)" if args.config == "car" else
                  "%s inference frames/sec (whole job)" % args.config,
        "value": round(fps, 3), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
        "trials": {"n": len(trial_dts), "steps_each": args.steps, "reported": "median",
                   "ms_per_step_min": round(min(trial_dts) / args.steps * 1e3, 4),
                   "ms_per_step_max": round(max(trial_dts) / args.steps * 1e3, 4)},
        "higher_is_better": True, "scaling": "weak",
        "vs_baseline": round(fps / world / PUBLISHED_FPS, 3) if headline else None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s, batch=%d, random-init SA-SSD weights, points resident in HBM" % (w["desc"], B),
                   "frames_per_step_per_gpu": B, "frames_in_flight": S, "spconv_cfg": args.spconv_cfg,
                   "rulebook_pyramid": args.pyramid,
                   "launch": "eager host launches" if args.eager else "one hipGraph replay per frame",
                   "frame_graph": ("one branch per frame in flight (rulebooks / anchor masks in front of the feature path on the "
                                   "frame's own stream); fps_sequential, latency, per-stage timings and roofline_sparse on a "
                                   "two-branch plan (one frame at a time: its coordinate work beside its first convolutions)"
                                   if one_branch else "two branches (coordinate side stream)"),
                   "hip_runtime_env": {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")},
                   "parallelism": "frame-sharded x%d, no collective" % world,
                   "vs_baseline_ref": "reference readme.md:2 '25 FPS' (hardware unstated), per-GPU fps / 25"},
        "fps_in_flight": round(fps, 3), "fps_sequential": round(B * 1e3 / seq_ms, 3),
        "latency_ms_sync_per_frame": round(lat_ms, 3), "frame_graph_ms": None if frame_ms is None else round(frame_ms, 4),
        "roofline": {"bound": "mfma", "kernel": kname,
                     "achieved": round(exec_tf * mfma_mult, 2), "peak": mfma_peak, "unit": "TFLOP/s",
                     "frac": round(exec_tf * mfma_mult / mfma_peak, 4),
                     "fp32_product_tflops": round(exec_tf, 2),          # the like-for-like figure across rounds
                     "algorithmic_tflops": round(conv_flops / (kernel_ms * 1e-3) / 1e12, 2),   # SURVEY 8(d) direct-conv flops / launch
                     "frac_of_fp32_mfma_peak": round(exec_tf / PEAK_F32_MFMA_TF, 4),
                     "ms_per_launch": round(kernel_ms, 4), "executed_flops_per_launch": exec_flops,
                     "layer": {"ms": round(conv_iso, 4), "direct_conv_flops": conv_flops,
                               "effective": round(eff_tf, 2), "effective_frac": round(eff_tf / PEAK_F32_MFMA_TF, 4),
                               "kernel_ms": {k: round(v, 4) for k, v in w4_parts.items()}},
                     "traffic": traffic, "traffic_measured_at": traffic_at,
                     "traffic_unit": "HBM-side bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc "
                                     "passes, profiles/)",
                     "note": "achieved/frac = flops EXECUTED on the MFMA pipe by the dominant kernel (Winograd F(4x4,3x3): "
                             "1/4 of the direct convolution) / its mean launch duration, HIP events on the launch stream "
                             "(= what rocprofv3 --kernel-trace reports, profiles/); layer.effective = the layer's direct-"
                             "convolution flops of SURVEY 8(d) over the whole layer (input transform + GEMM + output "
                             "transform).  Split geometry (default since round 4): every fp32 product is 8 bf16 piece "
                             "products on the bf16 MFMA, so achieved = 8 x fp32_product_tflops against the 2500 TF bf16 "
                             "peak; frac_of_fp32_mfma_peak prices the same launch against the 157.3 TF fp32-MFMA peak "
                             "the rounds 2-3 kernel ran on (--wino4-cfg 1 runs that kernel)"},
        "roofline_sparse": {"bound": "hbm", "kernels": "7 rulebooks (fused pyramid) + 14 sparse-conv launches, timed as "
                                                       "one hipGraph" if not args.eager else "eager, isolated pass",
                            "achieved": round(sp_gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                            "frac": round(sp_gbs / PEAK_HBM_GBS, 4),
                            "frac_of_measured_copy_peak": round(sp_gbs / MEASURED_HBM_GBS, 4),
                            "bytes_gs": work["bytes_gs"], "bytes_min": work["bytes_min"],
                            "rulebook_bytes": work["rulebook_bytes"], "flops": work["flops"],
                            "ms": round(sp_ms, 4), "ms_eager_isolated": round(iso_ms["sparse"], 4), "rows": work["n"],
                            "ms_convs_only": None if sp_convs_ms is None else round(sp_convs_ms, 4),
                            "ms_pyramid_only": None if sp_pyr_ms is None else round(sp_pyr_ms, 4),
                            "counters": sp_counters, "counters_measured_at": sp_counters_at,
                            "counters_unit": "per sparse-conv kernel: fractions of its wave-cycles parked (SQ_WAIT_ANY), issue-"
                                             "stalled (SQ_WAIT_INST_ANY), issuing (SQ_ACTIVE_INST_ANY); mfma_busy_fraction = "
                                             "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x SQ_BUSY_CYCLES per SE)",
                            "traffic": sp_traffic, "traffic_measured_at": sp_traffic_at,
                            "traffic_unit": "fabric-side bytes per pass, upper bound 2 x FETCH_SIZE + WRITE_SIZE over all "
                                            "rulebook / sparse-conv dispatches (separate rocprofv3 --pmc passes, profiles/): "
                                            "near bytes_min, the gathers of bytes_gs are served by the XCD L2s",
                            "note": "achieved = (bytes_gs + rulebook_bytes) / ms: the gather-scatter MODEL bytes of SURVEY "
                                    "8(d), which count every rulebook pair's row fetch -- an algorithmic rate, not an HBM "
                                    "rate (see traffic)"},
        "stage_ms": {k: round(v, 4) for k, v in sorted(iso_ms.items())},
        "bev_total_ms": round(bev_total_ms, 4),
        "detections_last_frame": ndet, "guided_anchor_candidates_last_frame": ncand,
    }
    out["csrc_hash"] = csrc_hash()
    # the inference measurement is complete: its plans, graphs and streams go before the other records of the line are taken (the
    # training record ran 238 instead of 276 samples/s with three captured frame graphs still alive in the process)
    del plans, plan                                        # (iso_plan stays: the parity record of cpu_baseline reads it)
    torch.cuda.empty_cache()
    if watchdog is not None:
        emit["line"] = out
        watchdog.start()
        try:
            if world == 1:                                       # (N > 1: a collective -- every rank measures in this process)
                try:
                    out["train"] = child_record(["--mode", "train", "--steps", str(args.train_steps), "--warmup",
                                                 str(args.train_warmup)], 400)
                except Exception as e:                           # fall back to the in-process measurement
                    out["train"] = train_measure(args, dev, rank, world, "car", "bf16", args.train_steps, args.train_warmup)
                    out["train"]["child_error"] = "%s: %s" % (type(e).__name__, e)
            else:
                out["train"] = train_measure(args, dev, rank, world, "car", "bf16", args.train_steps, args.train_warmup)
        except Exception as e:                                   # the inference line is still valid without it
            out["train"] = {"error": "%s: %s" % (type(e).__name__, e)}
        finally:
            watchdog.cancel()
    # ---- BASELINE configs[3] / [4] in the same line (N = 1 only: no collective, rank 0): short runs of the same measurement
    if world == 1 and headline and not args.no_extra and not args.eager:
        pyr = ["--pyramid", args.pyramid]
        keep = ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "trials", "config", "fps_sequential",
                "latency_ms_sync_per_frame", "roofline_sparse", "measured_by")
        for key, argv, fallback in (
                ("infer_multi", ["--config", "multi", "--steps", "60", "--warmup", "10", "--no-cpu-baseline", "--inflight", str(S)] + pyr,
                 lambda: infer_extra(dev, "multi", 60, 10, S, args.pyramid == "persistent")),
                ("infer_waymo", ["--config", "waymo", "--steps", "40", "--warmup", "8", "--no-cpu-baseline", "--inflight", str(S)] + pyr,
                 lambda: infer_extra(dev, "waymo", 40, 8, S, args.pyramid == "persistent")),
                ("train_waymo", ["--mode", "train", "--config", "waymo", "--steps", "12", "--warmup", "4", "--frames", "8"],
                 lambda: train_measure(args, dev, rank, world, "waymo", "bf16", 12, 4, frames=8))):
            t0 = time.perf_counter()
            try:
                try:
                    rec = child_record(argv, 300)
                    out[key] = rec if key == "train_waymo" else {k: rec[k] for k in keep if k in rec}
                except Exception as e:
                    out[key] = fallback()
                    out[key]["child_error"] = "%s: %s" % (type(e).__name__, e)
            except Exception as e:                               # the headline line is still valid without it
                out[key] = {"error": "%s: %s" % (type(e).__name__, e)}
            out[key]["wall_s"] = round(time.perf_counter() - t0, 1)
            torch.cuda.empty_cache()
    if world == 1 and headline and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(model, w, gpu_plan=iso_plan)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
