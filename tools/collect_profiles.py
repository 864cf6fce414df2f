#!/usr/bin/env python
"""Copy the evidence of tools/gpu_profiles.sh (gpurun_out/prof, scratch) into profiles/ (tracked): bench JSON lines,
rocprofv3 --kernel-trace --stats summaries, and the PMC (FETCH_SIZE / WRITE_SIZE) traffic records.
usage: python tools/collect_profiles.py [round tag, default r02]"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, DST = os.path.join(ROOT, "gpurun_out", "prof"), os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
sys.path.insert(0, ROOT)
import sassd  # noqa: E402,F401
from sassd import _C  # noqa: E402
CSRC = _C.csrc_hash()          # stamp: bench.py drops a traffic record measured on other kernel sources


def json_line(path):
    for line in open(path):
        if line.startswith("{"):
            return json.loads(line)
    return None


def counter(path, name, pat):
    if not os.path.exists(path):
        return 0, 0
    d = json.load(open(path))
    if "error" in d:
        return 0, 0
    tot = n = 0
    for k, v in d.items():
        if pat in k and name in v:
            tot += v[name]["mean_per_dispatch"] * v[name]["dispatches"]
            n += v[name]["dispatches"]
    return tot, n


for name in ("bench_inflight3", "bench_inflight1", "bench_multi", "bench_waymo", "bench_train", "bench_train_waymo_trace"):
    ks = os.path.join(SRC, name + "_kernel_stats.txt")
    if os.path.exists(ks):
        shutil.copy(ks, os.path.join(DST, "%s_%s_kernel_stats.txt" % (tag, name)))
    lg = os.path.join(SRC, name + "_under_rocprof.log")
    if os.path.exists(lg):
        d = json_line(lg)
        if d:
            json.dump(d, open(os.path.join(DST, "%s_%s_under_rocprof.json" % (tag, name)), "w"), indent=1)
for name in ("bench_default", "bench_train_bf16", "bench_train_fp32", "bench_train_waymo", "bench_train_forceddp",
             "bench_multi", "bench_waymo", "bench_20steps", "bench_fp32mfma"):
    lg = os.path.join(SRC, name + ".log")
    d = json_line(lg) if os.path.exists(lg) else None
    if d:
        json.dump(d, open(os.path.join(DST, "%s_%s.json" % (tag, name)), "w"), indent=1)
for name in ("spconv_layers_car", "spconv_layers_multi", "spconv_layers_waymo", "mfma4x4_probe", "wino4_geometries",
             "sparse_timeline_car_graph", "sparse_timeline_car_eager", "sparse_timeline_multi_graph", "full_tests_tail",
             "conv1x1_bf16_timing"):
    src = os.path.join(SRC, name + ".txt")
    if os.path.exists(src) and os.path.getsize(src) > 10:
        shutil.copy(src, os.path.join(DST, "%s_%s.txt" % (tag, name)))
bt = os.path.join(SRC, "bf16_conv_timing.json")
if os.path.exists(bt) and os.path.getsize(bt) > 10:
    shutil.copy(bt, os.path.join(DST, "%s_bf16_conv_timing.json" % tag))

# Winograd F(4x4) GEMM launch: HBM-side traffic per launch
f, nf = counter(os.path.join(SRC, "wino4_FETCH_SIZE.json"), "FETCH_SIZE", "wino4_gemm")
w, nw = counter(os.path.join(SRC, "wino4_WRITE_SIZE.json"), "WRITE_SIZE", "wino4_gemm")
parts = {}
for k in ("wino4_in", "wino4_out", "wino4_gemm"):
    a, na = counter(os.path.join(SRC, "wino4_FETCH_SIZE.json"), "FETCH_SIZE", k)
    b, nb = counter(os.path.join(SRC, "wino4_WRITE_SIZE.json"), "WRITE_SIZE", k)
    parts[k] = dict(FETCH_SIZE_kb_raw=a / max(na, 1), WRITE_SIZE_kb_raw=b / max(nb, 1))
Tp, cin, cout = 2304, 256, 256          # 2200 tiles padded to the 128-column block of the split geometry
alg = 36 * cin * Tp * 4 + 36 * cin * cout * 4 + 36 * cout * Tp * 4
rec = dict(
    csrc_hash=CSRC,
    kernel="wino4_gemm_kernel (36 GEMMs 256 x 256 x 2304 tiles of the BEV 256->256 3x3 layer @ 1x256x200x176, fp32 operands, "
           "products on the bf16 MFMA over split operands)",
    command="rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python tools/run_wino4.py --profile --reps 5 ; the same with "
            "--pmc WRITE_SIZE (two separate passes, tools/rocprof_pmc.py on each results.db)",
    FETCH_SIZE_kb_per_dispatch_raw=f / max(nf, 1), WRITE_SIZE_kb_per_dispatch_raw=w / max(nw, 1),
    correction="MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) "
               "coalesced reads -- both operands enter through 16 B/lane global->LDS DMA -- so fetch bytes = 2 x raw; "
               "WRITE_SIZE equals the product tensor (36 x 256 x 2304 x 4 B = 82 944 KiB) and is used as is; "
               "counters sit on the L2 fabric side, Infinity-Cache hits included",
    fetch_bytes_per_launch=int(2 * f / max(nf, 1) * 1024), write_bytes_per_launch=int(w / max(nw, 1) * 1024),
    algorithmic_bytes_per_launch=alg,
    per_kernel_raw_kb=parts)
rec["traffic_bytes_per_launch"] = rec["fetch_bytes_per_launch"] + rec["write_bytes_per_launch"]
rec["traffic_over_algorithmic"] = round(rec["traffic_bytes_per_launch"] / alg, 3)
if nf and nw:
    json.dump(rec, open(os.path.join(DST, "%s_wino4_gemm_hbm_traffic.json" % tag), "w"), indent=1)

# sparse segment: fabric-side traffic per pass (all rulebook + sparse-conv kernels)
for cfgname, reps in (("multi", 3), ("car", 5), ("waymo", 2)):
    fp, wp = os.path.join(SRC, "sparse_%s_FETCH_SIZE.json" % cfgname), os.path.join(SRC, "sparse_%s_WRITE_SIZE.json" % cfgname)
    if not (os.path.exists(fp) and os.path.exists(wp)):
        continue
    tf = sum(counter(fp, "FETCH_SIZE", pat)[0] for pat in ("spconv", "rb_", "hash_build", "pyr_"))
    tw = sum(counter(wp, "WRITE_SIZE", pat)[0] for pat in ("spconv", "rb_", "hash_build", "pyr_"))
    work = None
    for line in open(os.path.join(SRC, "sparse_%s_FETCH_SIZE.log" % cfgname)):
        if line.startswith("{'bytes_gs'"):
            work = eval(line)
    json.dump(dict(
        csrc_hash=CSRC,
        segment="7 rulebooks (fused pyramid) + 14 sparse convs, %s workload" % cfgname,
        command="rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python tools/run_sparse_only.py --config %s --reps %d ; the "
                "same with --pmc WRITE_SIZE" % (cfgname, reps),
        FETCH_SIZE_kb_per_pass_raw=tf / reps, WRITE_SIZE_kb_per_pass_raw=tw / reps,
        note="raw counter sums over every spconv_* / rb_* / hash_build dispatch of one pass; gathers are 16 B/lane loads "
             "(FETCH_SIZE under-reports them by up to 2x, MI355X_MICROARCH.md).  Even doubled, the fabric-side reads stay "
             "near the compulsory bytes (bytes_min) and far below the gather-scatter model bytes_gs the roofline_sparse "
             "fraction is quoted on: the gathers are served by the XCD L2s.",
        algorithmic=work), open(os.path.join(DST, "%s_sparse_%s_hbm_traffic.json" % (tag, cfgname)), "w"), indent=1)
# bf16 direct conv of the training step (B=2, 256->256, 200x176): fabric-side traffic per launch
f, nf = counter(os.path.join(SRC, "bf16conv_FETCH_SIZE.json"), "FETCH_SIZE", "conv2d_bf16_kernel")
w, nw = counter(os.path.join(SRC, "bf16conv_WRITE_SIZE.json"), "WRITE_SIZE", "conv2d_bf16_kernel")
if nf and nw:
    alg = 2 * 256 * 200 * 176 * 4 * 2 + 9 * 256 * 256 * 2           # fp32 map in + out, bf16 weights
    rec = dict(csrc_hash=CSRC,
               kernel="conv2d_bf16_kernel (3x3, 256->256, B=2 @200x176, bf16 MFMA operands, fp32 tensors in HBM)",
               command="rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python tools/run_bf16_conv.py --iters 5 ; the same with "
                       "--pmc WRITE_SIZE",
               FETCH_SIZE_kb_per_dispatch_raw=f / nf, WRITE_SIZE_kb_per_dispatch_raw=w / nw,
               correction="16 B/lane loads: fetch bytes = 2 x raw FETCH_SIZE (MI355X_MICROARCH.md, HBM section)",
               fetch_bytes_per_launch=int(2 * f / nf * 1024), write_bytes_per_launch=int(w / nw * 1024),
               algorithmic_bytes_per_launch=alg)
    rec["traffic_bytes_per_launch"] = rec["fetch_bytes_per_launch"] + rec["write_bytes_per_launch"]
    rec["traffic_over_algorithmic"] = round(rec["traffic_bytes_per_launch"] / alg, 3)
    json.dump(rec, open(os.path.join(DST, "%s_bf16_conv_hbm_traffic.json" % tag), "w"), indent=1)
# SQ stall counters of the three hot kernels (one pass each, eight SQ counters): where the wave-cycles go
stall = {}
for name, pats, suffix in (("stall_bf16conv", ("conv2d_bf16_kernel",), ""),
                           ("stall_sparse_car", ("spconv_gq_kernel<64, 64", "spconv_gs_kernel<32, 32", "spconv_gs_kernel<32, 64",
                                                 "spconv_pw_kernel", "spconv_gs_kernel", "spconv_gq_kernel"), "@car"),
                           ("stall_sparse_multi", ("spconv_gq_kernel<64, 64", "spconv_gs_kernel<32, 32", "spconv_gs_kernel<32, 64",
                                                   "spconv_pw_kernel", "spconv_gs_kernel", "spconv_gq_kernel"), "@multi"),
                           ("stall_sparse_car_r3geom", ("spconv_gs_kernel<64, 64",), "@car(round-3 geometry)"),
                           ("stall_wino4", ("wino4_gemm_kernel", "wino4_outin_kernel", "wino4_in_kernel", "wino4_out_kernel"), "")):
    path = os.path.join(SRC, name + "_SQ_WAVE_CYCLES.json")
    if not os.path.exists(path):
        continue
    d = json.load(open(path))
    if "error" in d:
        continue
    for pat in pats:
        acc, nd = {}, 0
        for k, v in d.items():
            if pat in k:
                for c, rec in v.items():
                    acc[c] = acc.get(c, 0.0) + rec["mean_per_dispatch"] * rec["dispatches"]
                    nd = max(nd, rec["dispatches"])
        if acc and acc.get("SQ_WAVE_CYCLES"):
            wc = acc["SQ_WAVE_CYCLES"]
            e = dict(dispatches=nd, per_dispatch={c: v / nd for c, v in acc.items()},
                     fraction_of_wave_cycles={c: round(v / wc, 4) for c, v in acc.items() if c != "SQ_WAVE_CYCLES"})
            # SQ_BUSY_CYCLES is summed over the 32 shader engines (8 XCDs x 4): / 32 = the kernel's cycles; the MFMA pipe
            # can be busy on 1024 SIMDs during each of them
            if acc.get("SQ_BUSY_CYCLES") and acc.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
                e["mfma_busy_fraction"] = round(acc["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * acc["SQ_BUSY_CYCLES"] / 32.0), 4)
            stall[pat + suffix] = e
if stall:
    json.dump(dict(csrc_hash=CSRC,
                   command="rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY "
                           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -- python tools/"
                           "run_bf16_conv.py | run_sparse_only.py --config car / multi [--spconv-cfg 10] | run_wino4.py --profile",
                   note="MI355X_MICROARCH.md: SQ_WAIT_ANY = wave parked (s_waitcnt / barrier), SQ_WAIT_INST_ANY = issue stall, "
                        "SQ_ACTIVE_INST_ANY = issuing; the three are disjoint and add up to ~SQ_WAVE_CYCLES (quad-cycles); "
                        "SQ_VALU_MFMA_BUSY_CYCLES counts cycles",
                   kernels=stall), open(os.path.join(DST, "%s_stall_breakdown.json" % tag), "w"), indent=1)
# LDS / clock counters of the bf16 conv (two passes): LDS-array cycles, bank conflicts, GRBM_GUI_ACTIVE (clock), MFMA busy
lds = {}
for fname in ("ldsclk_bf16conv_SQ_LDS_BANK_CONFLICT.json", "ldsclk_bf16conv_GRBM_GUI_ACTIVE.json"):
    path = os.path.join(SRC, fname)
    if not os.path.exists(path):
        continue
    d = json.load(open(path))
    if "error" in d:
        continue
    for k, v in d.items():
        if "conv2d_bf16_kernel" in k:
            for c, rec in v.items():
                lds[c] = rec["mean_per_dispatch"]
if lds:
    derived = {}
    if lds.get("GRBM_GUI_ACTIVE"):
        cyc = lds["GRBM_GUI_ACTIVE"] / 8.0                     # summed over the 8 XCDs
        derived["gpu_cycles_per_dispatch"] = cyc
        if lds.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            derived["mfma_busy_fraction"] = round(lds["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc, 4)   # 1024 SIMDs
    if lds.get("SQ_LDS_IDX_ACTIVE") and lds.get("SQ_LDS_BANK_CONFLICT") is not None:
        derived["lds_conflict_fraction_of_lds_cycles"] = round(lds["SQ_LDS_BANK_CONFLICT"] / lds["SQ_LDS_IDX_ACTIVE"], 4)
        if "gpu_cycles_per_dispatch" in derived:
            derived["lds_array_busy_fraction"] = round(lds["SQ_LDS_IDX_ACTIVE"] / 256.0 / derived["gpu_cycles_per_dispatch"], 4)
    json.dump(dict(csrc_hash=CSRC, kernel="conv2d_bf16_kernel<8, 4, 0>",
                   command="rocprofv3 --kernel-trace --pmc <counters> -- python tools/run_bf16_conv.py --only-fwd --iters 10 "
                           "(two passes of six counters)",
                   note="GRBM_GUI_ACTIVE is summed over the 8 XCDs; kernel duration x clock = cycles per dispatch; "
                        "SQ_VALU_MFMA_BUSY_CYCLES is summed over 1024 SIMDs, SQ_LDS_IDX_ACTIVE over 256 CUs",
                   per_dispatch=lds, derived=derived),
              open(os.path.join(DST, "%s_bf16_conv_lds_clock_counters.json" % tag), "w"), indent=1)
print("profiles written:", sorted(f for f in os.listdir(DST) if f.startswith(tag)))
