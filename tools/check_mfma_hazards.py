"""Static check of the hand-written MFMA code: hipcc pads no hazards for instructions inside inline asm,
so every v_mfma in the two-stage scan kernels must have, by construction, at least two wait states
between a VALU write (v_*, v_accvgpr_write) of one of its source registers and itself (CDNA3/4 ISA,
"VALU write VGPR -> MFMA read": 2 wait states; s_nop N supplies N + 1).

    python tools/check_mfma_hazards.py [file.s]     # without a file: compiles csrc/aura_knn.hip to asm first

Exit code 1 and a listing if a violation is found."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REG = re.compile(r"\b([va])(?:\[(\d+):(\d+)\]|(\d+))")


def regs(tok):
    """set of ('v'|'a', index) named by an operand token"""
    out = set()
    for m in REG.finditer(tok):
        f = m.group(1)
        if m.group(2) is not None:
            out.update((f, i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((f, int(m.group(4))))
    return out


def check(path, kernels=("coarse_scan_kernel",), need=2):
    bad, n_mfma, cur = [], 0, None
    window = []                                      # (wait states it supplies, written regs, text)
    for ln, line in enumerate(open(path), 1):
        t = line.split(";")[0].strip()
        if not t:
            continue
        if t.endswith(":"):
            if not t.startswith(".") and not t.startswith("$"):
                cur = t[:-1]
            window = []                              # new basic block: stay conservative only within one
            continue
        if t.startswith(".") or cur is None or not any(k in cur for k in kernels):
            continue
        parts = t.replace(",", " ").split()
        op = parts[0]
        if op.startswith("v_mfma"):
            n_mfma += 1
            srcs = set()
            for tok in parts[2:5]:
                srcs |= regs(tok)
            states = 0
            for st, wr, txt in reversed(window):
                if states >= need:
                    break
                if wr & srcs:
                    bad.append((cur, ln, txt, t))
                states += st
            window.append((1, set(), t))
            continue
        if op == "s_nop":
            window.append((int(parts[1], 0) + 1, set(), t))
        elif op.startswith("v_") and not op.startswith("v_nop") and not op.startswith("v_cmp"):
            window.append((1, regs(parts[1]) if len(parts) > 1 else set(), t))
        else:
            window.append((1, set(), t))
        window = window[-8:]
    return n_mfma, bad


def main():
    if len(sys.argv) > 1:
        path = sys.argv[1]
    else:
        path = os.path.join(tempfile.mkdtemp(prefix="aura_haz_"), "aura_knn.s")
        src = os.path.join(ROOT, "aura_snn_rag_amd", "csrc", "aura_knn.hip")
        subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950",
                               "-Wno-unused-function", "--cuda-device-only", "-S", src, "-o", path],
                              stderr=subprocess.DEVNULL)
    n, bad = check(path)
    print(f"{n} MFMA instructions checked in the scan kernels, {len(bad)} hazard(s)")
    for k, ln, w, m in bad[:20]:
        print(f"  {k[:60]} line {ln}: `{w}` too close to `{m}`")
    return 1 if bad or n == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
