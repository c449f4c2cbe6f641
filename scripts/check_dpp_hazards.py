"""Two checks of the fp64 kernels' device assembly (CPU: hipcc cross-compiles; tests/test_host_cpu.py runs them).
1. The hand-written DPP instructions (tiles64.h fmac64_row) sit in inline asm, where the compiler's hazard recogniser cannot see
that operand 1 goes through the DPP path: a VALU write of that register needs 2 wait states before the DPP read (and an EXEC
write 5).  This compiles the fp64 kernels to assembly and checks every v_*_dpp against the instructions in front of it.
2. The one-launch kernel's shape: batch64_step_kernel must stay at <= 420 registers with its steady K loop in ONE basic block of
128 MFMAs -- round 6 saw a 32-register excursion elsewhere in the kernel make the compiler break that loop into pieces (17.8 us per
K block instead of 14.8 on every tile: DESIGN 4.7).
    python scripts/check_dpp_hazards.py"""
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
        text = open(out).read()
        lines = [l.strip() for l in text.splitlines() if l.startswith("\t") and not l.strip().startswith((";", "."))]
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
    return bad, n, text


def kernel_shape(text, kernel_prefix):
    """(registers, MFMA count of the fullest basic block that has no AGPR copies) per kernel whose mangled name starts with the prefix"""
    out = {}
    for name, vg in re.findall(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", text):
        if not name.startswith(kernel_prefix): continue
        body = text[text.index("\n" + name + ":"):]
        body = body[:body.index(".end_amdhsa_kernel")] if ".end_amdhsa_kernel" in body else body
        best = 0
        for blk in re.split(r"\n\.LBB", body):
            m = len(re.findall(r"\bv_mfma_f64", blk))
            if m > best and "v_accvgpr_read" not in blk and m <= 128: best = m
        out[name] = (int(vg), best)
    return out

if __name__ == "__main__":
    rc = 0
    for src in ("chol64.hip", "batch64_step.hip", "mll64.hip"):
        bad, n, text = check(src)
        rc |= bad != 0
        if src == "batch64_step.hip":
            for k, (vg, mf) in kernel_shape(text, "_ZN4volt19batch64_step_kernel").items():
                print(f"  {k}: {vg} registers, steady K loop {mf} MFMAs in one block")
                rc |= vg > 420 or mf != 128
    sys.exit(rc)
