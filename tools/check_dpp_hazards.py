"""Static check of the hand-kept hazards around the inline-asm DPP instructions of dense_inl.h (hipcc does not look into asm
statements): on gfx940+ a VGPR written by a VALU instruction must not be read through DPP by either of the next two
instructions, and the result of a transcendental must not be read by the next instruction.  A label inside the window
counts as a violation (what ran before a branch target is unknown).  Run by `make -C 3dbodyanimation_amd/csrc hazards` on the
ISA of the shipped flags (part of `all`), and by tests/test_abi.py."""
import re
import sys


def regs(tok):
    m = re.match(r"-?v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"-?v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def main(path):
    ins = []
    for ln in open(path):
        t = ln.strip()
        if t.endswith(":") and not t.startswith(";"):
            ins.append("<label>")          # a branch target: what ran before it is unknown
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.startswith("//"):
            continue
        t = t.split(";")[0].strip()
        if t:
            ins.append(t)
    bad = 0
    n_dpp = 0
    for i, t in enumerate(ins):
        op = t.split()[0]
        if op == "<label>":
            continue
        ops = [o.strip() for o in t[len(op):].split(",")]
        if op in ("v_fmac_f64_dpp", "v_mov_b64_dpp"):
            n_dpp += 1
            src = regs(ops[1].split()[0])
            need = 2
        elif i > 0 and ins[i - 1].split()[0] in ("v_rsq_f64", "v_rsq_f64_e32", "v_rcp_f64_e32", "v_sqrt_f64_e32") and op.startswith("v_"):
            # (a transcendental that ends a block and is read at a branch target: hipcc's own hazard recogniser covers it,
            #  both instructions being compiler-visible; the asm-only case is the DPP read below)
            p = ins[i - 1]
            pops = [o.strip() for o in p[len(p.split()[0]):].split(",")]
            src = regs(pops[0])
            used = set()
            for o in ops[1:]:
                used |= regs(o.split()[0])
            if src & used:
                print("TRANS hazard:", p, "->", t)
                bad += 1
            continue
        else:
            continue
        ws = 0
        j = i - 1
        while j >= 0 and ws < need:
            q = ins[j]
            qop = q.split()[0]
            if qop == "<label>":
                # the wait states must be satisfied inside the block: a predecessor that jumps here is not visible
                print(f"DPP hazard (branch target {ws} wait states before):", t)
                bad += 1
                break
            if qop == "s_nop":
                ws += int(q.split()[1]) + 1
            else:
                if qop.startswith("v_") and not qop.startswith("v_cmp"):
                    qops = [o.strip() for o in q[len(qop):].split(",")]
                    if regs(qops[0].split()[0]) & src:
                        print(f"DPP hazard ({ws} wait states):", q, "->", t)
                        bad += 1
                        break
                ws += 1
            j -= 1
    print(f"{path}: {n_dpp} DPP instructions checked, {bad} violations")
    return bad


if __name__ == "__main__":
    sys.exit(1 if sum(main(p) for p in sys.argv[1:]) else 0)
