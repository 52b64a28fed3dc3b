#!/usr/bin/env python3
"""Per-kernel fingerprint (md5 of the disassembly, addresses stripped of their section base) of the gfx950 code objects in a
built library.  Used to show that a source refactor leaves the PRODUCT kernels' machine code unchanged (round 6: the
checked-accessor macros of the -DOVRFSR_BOUNDS build expand to the old pointer expressions in the product build).

    tools/isa_fingerprint.py [lib.so] > fp.json          tools/isa_fingerprint.py --diff a.json b.json
"""
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_costs  # noqa: E402


def fingerprint(lib):
    tmp = tempfile.mkdtemp(prefix="ovrfsr_fp_")
    out = {}
    try:
        for elf in isa_costs.extract_code_objects(lib, tmp):
            text = subprocess.run([isa_costs._tool("llvm-objdump"), "-d", "--no-show-raw-insn", "--no-leading-addr", elf],
                                  capture_output=True, text=True, check=True).stdout
            name, body = None, []
            for line in text.splitlines():
                m = re.match(r"^[0-9a-f]* ?<(.+)>:$", line.strip())
                if m:
                    if name:
                        out[name] = body
                    name, body = m.group(1), []
                elif name is not None and line.strip():
                    # branch targets are printed as absolute addresses + <sym+off>: keep the symbolic part only
                    body.append(re.sub(r"\b0x[0-9a-f]+ (<[^>]+>)", r"\1", re.sub(r"//.*$", "", line).strip()))
            if name:
                out[name] = body
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    names = isa_costs.demangle(list(out))
    return {names[k]: {"n": len(v), "md5": hashlib.md5("\n".join(v).encode()).hexdigest()} for k, v in out.items()}


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--diff":
        a, b = (json.load(open(p)) for p in sys.argv[2:4])
        bad = 0
        for k in sorted(set(a) | set(b)):
            if a.get(k) != b.get(k):
                bad += 1
                print("DIFF", k[:160], a.get(k), b.get(k))
        print("%d kernels/functions compared, %d differ" % (len(set(a) | set(b)), bad))
        return 1 if bad else 0
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "openvr_fsr_amd", "libopenvr_fsr_amd.so")
    json.dump(fingerprint(lib), sys.stdout, indent=0, sort_keys=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
