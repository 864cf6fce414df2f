"""Instruction mix of a compiled kernel's hottest loop, from the gfx950 disassembly -- a GPU-less proxy for the PMC finding
that VALU instructions never co-execute with MFMAs on this part (SQ_VALU_MFMA_COEXEC_CYCLES = 0, profiles/README.md): every
VALU op in an MFMA loop costs MFMA issue slots, so VALU-per-MFMA is the number to drive down before going to the GPU.

    python tools/isa_stats.py conv2d_wino            # all kernels of sa-ssd_amd/csrc/conv2d_wino.hip
    python tools/isa_stats.py conv2d_wino wino_fwd   # only kernels whose (mangled) name contains 'wino_fwd'

For each kernel: registers / LDS / scratch from the metadata, then the instruction mix of (a) the whole kernel and (b) the
innermost backward-branch loop that contains the most MFMAs (or the most instructions when there are none)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("v_accvgpr"):
        return "acc_mov"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def device_elf(obj, tmp):
    """the gfx950 code object embedded in a hipcc host object (.hip_fatbin section = a clang offload bundle)"""
    fat, out = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.o")
    subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj], check=True)
    r = subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + out], capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(out) or os.path.getsize(out) == 0:
        raise SystemExit("could not extract the gfx950 image from %s:\n%s" % (obj, r.stderr))
    return out


def kernels(elf):
    txt = subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", elf], capture_output=True, text=True).stdout
    cur, out = None, collections.OrderedDict()
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):(.*)$", line)
        if m and cur is not None:
            t = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", m.group(4))      # branch target as symbol + offset
            out[cur].append((int(m.group(3), 16), m.group(1), m.group(2), None if t is None else int(t.group(1), 16)))
    return out


def hottest_loop(ins):
    """(start, end) instruction indices of the backward branch span holding the most MFMAs (ties: most instructions,
    innermost)."""
    addr_to_i = {x[0]: i for i, x in enumerate(ins)}
    base, best = ins[0][0], None
    for i, (a, op, args, off) in enumerate(ins):
        if not op.startswith(("s_cbranch", "s_branch")) or off is None:
            continue
        tgt = base + off
        if tgt not in addr_to_i or tgt > a:
            continue
        j = addr_to_i[tgt]
        body = ins[j:i + 1]
        key = (sum(1 for x in body if classify(x[1]) == "mfma"), -len(body))
        if best is None or key > best[0]:
            best = (key, j, i)
    return None if best is None else best[1:]


def mix(ins):
    c = collections.Counter(classify(x[1]) for x in ins)
    return c, len(ins)


def fmt(c, n):
    keys = ["mfma", "valu", "acc_mov", "lds", "vmem", "salu", "waitcnt", "barrier", "branch", "other"]
    s = "  ".join("%s %d" % (k, c[k]) for k in keys if c[k])
    if c["mfma"]:
        s += "   | VALU/MFMA %.2f  LDS/MFMA %.2f" % ((c["valu"] + c["acc_mov"]) / c["mfma"], c["lds"] / c["mfma"])
    return "%5d instr: %s" % (n, s)


def top_valu(ins, k=8):
    c = collections.Counter(x[1] for x in ins if classify(x[1]) in ("valu", "acc_mov"))
    return ", ".join("%s x%d" % kv for kv in c.most_common(k))


def metadata(elf):
    txt = subprocess.run([LLVM + "/llvm-readelf", "--notes", elf], capture_output=True, text=True).stdout
    out = {}
    for blk in re.split(r"\n\s*- \.agpr_count:", "\n" + txt)[1:]:
        blk = ".agpr_count:" + blk
        name = re.search(r"\.name:\s*(\S+)", blk)
        if not name:
            continue
        g = lambda key: (re.search(r"\." + key + r":\s*(\d+)", blk) or [None, "?"])[1]
        out[name.group(1)] = dict(vgpr=g("vgpr_count"), agpr=g("agpr_count"), sgpr=g("sgpr_count"),
                                  lds=g("group_segment_fixed_size"), scratch=g("private_segment_fixed_size"),
                                  spill=g("vgpr_spill_count"))
    return out


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    obj = os.path.join(ROOT, "sa-ssd_amd", "lib", "obj", sys.argv[1] + ".o")
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    with tempfile.TemporaryDirectory() as tmp:
        elf = device_elf(obj, tmp)
        meta = metadata(elf)
        for name, ins in kernels(elf).items():
            if pat not in name or not ins:
                continue
            m = meta.get(name, {})
            print("== %s" % name)
            if m:
                print("   vgpr %s  agpr %s  sgpr %s  lds %s B  scratch %s B  spills %s" % (
                    m["vgpr"], m["agpr"], m["sgpr"], m["lds"], m["scratch"], m["spill"]))
            print("   kernel " + fmt(*mix(ins)))
            loop = hottest_loop(ins)
            if loop:
                body = ins[loop[0]:loop[1] + 1]
                print("   loop   " + fmt(*mix(body)))
                print("   loop VALU: " + top_valu(body))


if __name__ == "__main__":
    main()
