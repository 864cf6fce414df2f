"""The bf16 conv issues global loads by hand (inline asm + explicit s_waitcnt); between such a load and the wait that covers
it no compiler-generated instruction may read, copy or spill the destination registers (the compiler does not know the
load is in flight).  This compiles the kernel's device assembly and runs tools/check_inflight_regs.py over the shipped
instantiations -- a guard against a compiler or source change re-introducing the hazard DESIGN.md section 4 describes."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_bf16_conv_has_no_instruction_on_in_flight_registers(tmp_path):
    src = os.path.join(ROOT, "sa-ssd_amd", "csrc", "conv2d_bf16.hip")
    asm = str(tmp_path / "conv2d_bf16.s")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math",
             "-fhip-fp32-correctly-rounded-divide-sqrt", "-ffp-contract=on"]
    subprocess.run([HIPCC] + flags + ["-S", "--cuda-device-only", src, "-o", asm], check=True, cwd=os.path.dirname(src),
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for sym in ("conv2d_bf16_kernelILi8ELi4ELi0E", "conv2d_bf16_kernelILi4ELi4ELi0E"):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_inflight_regs.py"), asm, sym],
                             check=True, capture_output=True, text=True).stdout
        assert out.strip().endswith("flags: 0"), out[-2000:]
