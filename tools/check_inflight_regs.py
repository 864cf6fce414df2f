"""ISA check for the kernels that issue global loads by hand (inline asm + explicit s_waitcnt): between the load and the
s_waitcnt that covers it, no compiler-generated instruction may touch the destination registers (a copy or a spill there
moves stale data).  Linear scan of the device assembly of one kernel; flags are to be read by a human (the scan does not
follow branches: after an unconditional branch it starts clean).   python tools/check_inflight_regs.py /tmp/bf.s <kernel symbol substring>"""
import re
import sys


def regs_of(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def main():
    text = open(sys.argv[1]).read()
    sym = sys.argv[2]
    m = re.search(r"^(\S*%s\S*):[^\n]*\n" % re.escape(sym), text, re.M)
    body = text[m.end():text.index(".end_amdhsa_kernel", m.end())].split("\n")
    inflight = []          # FIFO of register sets
    in_asm = False
    flags = 0
    for ln, line in enumerate(body):
        l = line.strip()
        if l.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if l.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not l or l.startswith(";") or l.startswith("."):
            continue
        if l.startswith("s_branch") or l.startswith("s_endpgm") or l.startswith("s_setpc"):
            inflight.clear()       # what follows is reached from elsewhere: state unknown, start clean
            continue
        toks = re.findall(r"v\[\d+:\d+\]|v\d+", l)
        if in_asm:
            if l.startswith("global_load"):
                inflight.append(regs_of(toks[0]))
            elif l.startswith("s_waitcnt"):
                n = int(re.search(r"vmcnt\((\d+)\)", l).group(1)) if "vmcnt" in l else None
                if n is not None:
                    while len(inflight) > n:
                        inflight.pop(0)
            continue
        if l.startswith("s_waitcnt") and "vmcnt(0)" in l:
            inflight.clear()
            continue
        used = set()
        for t in toks:
            used |= regs_of(t)
        hot = set().union(*inflight) if inflight else set()
        if used & hot:
            flags += 1
            print("line %d: %s   <- in flight: %s" % (ln, l, sorted(used & hot)))
    print("flags:", flags)


if __name__ == "__main__":
    main()
