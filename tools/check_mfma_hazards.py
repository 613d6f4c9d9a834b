"""Static check of the hand-written MFMA code: hipcc pads no hazards for instructions inside inline asm,
so every v_mfma in the two-stage scan kernels must have, by construction, at least two wait states
between a VALU write (v_*, v_accvgpr_write) of one of its source registers and itself (CDNA3/4 ISA,
"VALU write VGPR -> MFMA read": 2 wait states; s_nop N supplies N + 1), and every read of an MFMA's
result by another instruction (v_accvgpr_read, ...) must come at least 11 wait states behind it.  The
look-back follows fall-through and branch edges (loop back edges included), not only straight-line code.

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


BRANCH = re.compile(r"^s_c?branch\w*\s+(\.\w+)")
ENDS_FLOW = ("s_branch", "s_endpgm", "s_setpc_b64", "s_swappc_b64")


def _entry(op, parts, t):
    """(wait states the instruction supplies, registers it writes as a VALU op, text)"""
    if op == "s_nop":
        return (int(parts[1], 0) + 1, set(), t)
    if op.startswith("v_") and not op.startswith(("v_nop", "v_cmp", "v_mfma")):
        return (1, regs(parts[1]) if len(parts) > 1 else set(), t)
    return (1, set(), t)


def check(path, kernels=("coarse_scan_kernel",), need=2, need_read=11):
    """Two hazards hipcc does not pad inside inline asm:
      * VALU write of a register -> MFMA reading it as A/B/C: `need` wait states;
      * MFMA writing its accumulators -> a non-MFMA instruction reading them: `need_read` wait states
        (v_mfma_f32_16x16x32_bf16: 8 passes; the kernels put 13 behind a tile's last MFMAs).
    The look-back does not stop at labels: a block inherits the tail of the code that falls into it and
    of every branch that targets it (back edges of the tile loop included)."""
    funcs, cur = {}, None
    for ln, line in enumerate(open(path), 1):
        t = line.split(";")[0].strip()
        if not t:
            continue
        if t.endswith(":"):
            lab = t[:-1]
            if not lab.startswith(".") and not lab.startswith("$"):
                cur = lab
                funcs[cur] = []
            elif cur is not None:
                funcs[cur].append(("label", lab, ln))
            continue
        if t.startswith(".") or cur is None:
            continue
        funcs[cur].append(("ins", t, ln))
    bad, n_mfma = [], 0
    for fn, items in funcs.items():
        if not any(k in fn for k in kernels):
            continue
        # predecessors of every label: the code in front of it (unless that ends the flow) and each branch site
        label_at = {lab: i for i, (kind, lab, _) in enumerate(items) if kind == "label"}
        preds = {lab: [] for lab in label_at}
        for i, (kind, t, _) in enumerate(items):
            if kind == "ins":
                m = BRANCH.match(t)
                if m and m.group(1) in preds:
                    preds[m.group(1)].append(i)              # look back from the branch instruction itself
        for lab, i in label_at.items():
            j = i - 1
            while j >= 0 and items[j][0] == "label":
                j -= 1
            if j >= 0 and not items[j][1].split()[0].startswith(ENDS_FLOW):
                preds[lab].append(j)

        def lookback(i, budget, depth=0):
            """yield (states accumulated before it, entry) for the instructions that can precede position i"""
            states, j = 0, i - 1
            while j >= 0 and states < budget:
                kind, t, _ = items[j]
                if kind == "label":
                    if depth < 3:
                        for p_ in preds.get(t, []):
                            if p_ == j - 1:
                                continue                     # the fall-through predecessor is walked below
                            for st, e in lookback(p_ + 1, budget - states, depth + 1):
                                yield states + st, e
                    j -= 1
                    continue
                parts = t.replace(",", " ").split()
                e = _entry(parts[0], parts, t)
                yield states, (e, parts)
                states += e[0]
                if parts[0].startswith(ENDS_FLOW) and j != i - 1:
                    return
                j -= 1

        for i, (kind, t, ln) in enumerate(items):
            if kind != "ins":
                continue
            parts = t.replace(",", " ").split()
            op = parts[0]
            if op.startswith("v_mfma"):
                n_mfma += 1
                srcs = set()
                for tok in parts[2:5]:
                    srcs |= regs(tok)
                for st, (e, _) in lookback(i, need):
                    if st < need and e[1] & srcs:
                        bad.append((fn, ln, e[2], t))
            elif op.startswith(("v_", "ds_write", "global_store", "buffer_store", "scratch_store")) and len(parts) > 1:
                reads = set()
                for tok in (parts[2:] if op.startswith("v_") else parts[1:]):
                    reads |= regs(tok)
                if not reads:
                    continue
                for st, (e, pp) in lookback(i, need_read):
                    if st < need_read and pp[0].startswith("v_mfma") and regs(pp[1]) & reads:
                        bad.append((fn, ln, e[2], t))
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
