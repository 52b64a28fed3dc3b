#!/usr/bin/env python3
"""tools/isa_peephole.py IN.s OUT.s -- build step between the compiler and the assembler for the gfx950 kernel TUs.

One rewrite, for one measured property of the MI355X VALU that the compiler's cost model does not know
(profiles/r05_cndmask_vop2.txt, tools/ubench/valu_rates.hip):

    v_cndmask_b32_e32 vD, vA, vB, vcc      VOP2 encoding, VCC read implicitly
 -> v_cndmask_b32_e64 vD, vA, vB, vcc      VOP3 encoding, the same VCC as an explicit SGPR-pair operand

Back-to-back VOP2 selects on VCC issue at ~23.7 cycles each instead of 4.1 (a compare followed by five selects -- NVScaler's
`sh ? t1 : t0` groups -- costs 97 cycles against 25 with the mask in an SGPR pair); the VOP3 form of the very same select runs
at 4.2 even on VCC.  The result is bit-identical: same operation, same operands, 4 more bytes of code.  LLVM's
SIShrinkInstructions turns every VOP3 select whose mask landed in VCC into the VOP2 form and has no switch.
Only plain `v_cndmask_b32_e32 ..., vcc` lines are touched (no DPP / SDWA forms); everything else passes through.
Prints how many selects it widened."""
import re
import sys

PAT = re.compile(r"^(\s*)v_cndmask_b32_e32(\s+[^;\n]*?),\s*vcc(\s*(?:;.*)?)$")


def rewrite(text):
    n = 0
    out = []
    for line in text.split("\n"):
        m = PAT.match(line)
        if m and "dpp" not in line and "sdwa" not in line:
            line = "%sv_cndmask_b32_e64%s, vcc%s" % (m.group(1), m.group(2), m.group(3))
            n += 1
        out.append(line)
    return "\n".join(out), n


def main():
    src, dst = sys.argv[1], sys.argv[2]
    text, n = rewrite(open(src).read())
    open(dst, "w").write(text)
    print("isa_peephole: %d v_cndmask_b32 VOP2 -> VOP3 in %s" % (n, src))


if __name__ == "__main__":
    main()
