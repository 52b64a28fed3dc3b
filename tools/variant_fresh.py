#!/usr/bin/env python3
"""tools/variant_fresh.py NAME [FLAGS] -> exit 0 when ab/NAME.so carries a stamp equal to the hash of (FLAGS, the sources it is built from);
--stamp NAME FLAGS writes the stamp (tools/build_variant.sh, after a successful build)."""
import glob
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_hash(flags):
    h = hashlib.sha256(flags.encode())
    files = sorted(glob.glob(os.path.join(ROOT, "openvr_fsr_amd", "csrc", "*.*")) + [os.path.join(ROOT, "include", "openvr_fsr_amd.h"),
                                                                                  os.path.join(ROOT, "openvr_fsr_amd", "csrc", "Makefile")])
    for f in files:
        if os.path.isfile(f) and not f.endswith((".o", ".so")):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()


def stamp_path(name):
    return os.path.join(ROOT, "ab", name + ".so.stamp")


def fresh(name, flags):
    lib = os.path.join(ROOT, "ab", name + ".so")
    try:
        return os.path.exists(lib) and open(stamp_path(name)).read().strip() == source_hash(flags)
    except OSError:
        return False


if __name__ == "__main__":
    if sys.argv[1] == "--stamp":
        open(stamp_path(sys.argv[2]), "w").write(source_hash(sys.argv[3] if len(sys.argv) > 3 else "") + "\n")
        sys.exit(0)
    sys.exit(0 if fresh(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "") else 1)
