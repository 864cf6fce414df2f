"""Times the bf16 BEV training kernels next to the fp32 ones at the bench shape (B=2, 256->256, 200x176) with
torch events on the current stream:  python tools/run_bf16_conv.py [--iters 20]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sassd  # noqa: E402,F401
from sassd import kernels as K  # noqa: E402


def timed(fn, iters):
    for _ in range(max(10, iters // 2)):     # clocks settle over the first dozens of launches
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--ablate", action="store_true", help="time the bf16 conv with parts switched off")
    ap.add_argument("--only-fwd", action="store_true", help="only the bf16 forward conv (for counter passes)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    x = torch.randn(2, 256, 200, 176, device=dev)
    dy = torch.randn(2, 256, 200, 176, device=dev) * 0.01
    out = {}
    if a.only_fwd:
        w = torch.randn(256, 256, 3, 3, device=dev) / 48
        pk = K.conv2d_bf16_pack_weight(w)
        out["fwd3x3_bf16_ms"] = timed(lambda: K.conv2d_bf16_fwd(x, pk, 256), a.iters)
        print(json.dumps(out, indent=1))
        return
    out["wgrad3x3_fp32_ms"] = timed(lambda: K.conv2d_bwd_weight(x, dy, 3), a.iters)
    out["wgrad3x3_bf16_ms"] = timed(lambda: K.conv2d_bwd_weight(x, dy, 3, bf16=True), a.iters)
    out["wgrad1x1_fp32_ms"] = timed(lambda: K.conv2d_bwd_weight(x, dy, 1), a.iters)
    out["wgrad1x1_bf16_ms"] = timed(lambda: K.conv2d_bwd_weight(x, dy, 1, bf16=True), a.iters)
    flops = 2.0 * 2 * 200 * 176 * 256 * 256 * 9
    out["wgrad3x3_bf16_tflops"] = flops / out["wgrad3x3_bf16_ms"] / 1e9
    out["wgrad3x3_fp32_tflops"] = flops / out["wgrad3x3_fp32_ms"] / 1e9
    if hasattr(K, "conv2d_bf16_fwd"):
        w = torch.randn(256, 256, 3, 3, device=dev) / 48
        pk = K.conv2d_bf16_pack_weight(w)
        p4 = K.conv2d_wino4_pack_weight(w)
        out["fwd3x3_bf16_ms"] = timed(lambda: K.conv2d_bf16_fwd(x, pk, 256), a.iters)
        out["fwd3x3_wino4_fp32_ms"] = timed(lambda: K.conv2d_wino4_fwd(x, p4, 256, None, None), a.iters)
        out["fwd3x3_bf16_tflops"] = flops / out["fwd3x3_bf16_ms"] / 1e9
        if a.ablate:
            names = {1: "no_input_staging", 2: "hot_weights", 4: "no_mfma", 8: "no_stores", 3: "no_global_loads",
                     15: "only_lds", 11: "mfma_and_lds_only", 16: "input_hot_in_l2",
                     43: "mma_loop_no_b_reads", 75: "mma_loop_no_barriers",
                     171: "mma_loop_mfma_and_barriers_only", 235: "mma_loop_mfma_only",
                     172: "loaders_only", 164: "loaders_and_stores_only",
                     168: "staging_and_mfma_only", 40: "staging_mfma_weights_no_b_reads",
                     0x10000: "mma_waves_default_priority", 0x20000: "loader_waves_high_priority",
                     0x40000: "loader_quad_fastest_order_r02", 0: "default_again"}
            for flag, name in names.items():
                out["ablate_" + name + "_ms"] = timed(lambda: K.conv2d_bf16_fwd(x, pk, 256, cfg=flag), a.iters)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
