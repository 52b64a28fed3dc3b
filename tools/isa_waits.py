#!/usr/bin/env python3
"""tools/isa_waits.py <file.s>... -- find vector-memory loads that the scheduler serialised.

For every basic block of every kernel in a `hipcc -S --cuda-device-only` listing: if an `s_waitcnt vmcnt(N)` sits BETWEEN two
global/buffer loads of the block, print the kernel, the block, N and how many of the block's loads had been issued by then.
"wait vmcnt(1) after 2 of 8 loads" means loads 3..8 are issued only after load 1 has returned: the memory-level parallelism the
source asked for (all loads, then a fence, then the arithmetic) was lost to a use of an early result that the scheduler hoisted
between the loads.  Round 3 found C5's outside-tile kernel in exactly that state (decode written next to each load; fix: keep
the loaded words raw across the fence, decode after it: +4 % on C5, 72 -> 61 VGPRs; profiles/r03_outside_raw_fence.txt).

Listing:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -fno-slp-vectorize -S --cuda-device-only \
              -Iopenvr_fsr_amd/csrc -Iinclude openvr_fsr_amd/csrc/fsr_kernels.hip -o /tmp/fsr.s
Loops with one load per iteration are reported too (wait after 1 of 1 is not: a block needs two loads); read the block.
"""
import re
import sys


def scan(path):
    kern, blk, seq, out = None, None, [], []

    def flush():
        loads = [i for i, (t, _) in enumerate(seq) if t == "L"]
        if len(loads) >= 2:
            for i, (t, v) in enumerate(seq):
                if t == "W" and loads[0] < i < loads[-1]:
                    out.append((kern, blk, v, sum(1 for j in loads if j < i), len(loads)))
                    break
        del seq[:]

    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            flush(); kern, blk = m.group(1), "entry"; continue
        m = re.match(r"^(\.LBB\w+):", line)
        if m:
            flush(); blk = m.group(1); continue
        if kern is None:
            continue
        if re.search(r"\b(global_load|buffer_load)", line):
            seq.append(("L", None))
        m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", line)
        if m:
            seq.append(("W", m.group(1)))
        if "s_endpgm" in line:
            flush(); kern = None
    return out


def main():
    if len(sys.argv) < 2:
        print(__doc__); return 2
    for path in sys.argv[1:]:
        for kern, blk, n, before, total in scan(path):
            print("%-100s %-12s wait vmcnt(%s) after %d of %d loads" % (kern[:100], blk, n, before, total))
    return 0


if __name__ == "__main__":
    sys.exit(main())
