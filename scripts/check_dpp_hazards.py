"""The hand-written DPP instructions (tiles64.h fmac64_row) sit in inline asm, where the compiler's hazard recogniser cannot see
that operand 1 goes through the DPP path: a VALU write of that register needs 2 wait states before the DPP read (and an EXEC
write 5).  This compiles the fp64 kernels to assembly and checks every v_*_dpp against the instructions in front of it.
    python scripts/check_dpp_hazards.py            (CPU: hipcc cross-compiles)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from volt_amd.build import CSRC, FLAGS, _hipcc

def regs(tok):
    m = re.match(r"-?\|?v\[(\d+):(\d+)\]", tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"-?\|?v(\d+)", tok)
    return {int(m.group(1))} if m else set()

def check(src):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run([_hipcc(), *[f for f in FLAGS if f != "-fPIC"], '-DVOLT_SOURCE_HASH="x"', "--cuda-device-only", "-S", "-o", out,
                        os.path.join(CSRC, src)], check=True, stderr=subprocess.DEVNULL)
        lines = [l.strip() for l in open(out) if l.startswith("\t") and not l.strip().startswith((";", "."))]
    bad = n = 0
    for i, l in enumerate(lines):
        op = l.split()[0]
        if not op.endswith("_dpp"): continue
        n += 1
        src0 = regs(l.split(None, 1)[1].split(",")[1].strip())
        states = 0
        for p in reversed(lines[max(0, i - 6):i]):
            pop = p.split()[0]
            if pop == "s_nop": states += int(p.split()[1]) + 1; continue
            if states < 2 and pop.startswith("v_") and not pop.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
                if regs(p.split(None, 1)[1].split(",")[0].strip()) & src0:
                    bad += 1; print(f"{src}: VALU write {states} wait state(s) ahead of its DPP read:\n    {p}\n    {l}")
            if states < 5 and "exec" in p.split(None, 1)[-1].split(",")[0] and pop.startswith(("s_", "v_cmpx")):
                bad += 1; print(f"{src}: EXEC write {states} wait state(s) ahead of a DPP instruction:\n    {p}\n    {l}")
            states += 1
            if states >= 5: break
    print(f"{src}: {n} DPP instructions, {bad} hazards")
    return bad

if __name__ == "__main__":
    sys.exit(1 if sum(check(s) for s in ("chol64.hip", "batch64_step.hip", "mll64.hip")) else 0)
