#!/bin/bash
OVRFSR_LIB=$PWD/ab/audit.so timeout 1500 python tools/debug/tie_audit.py 1.0 > gpurun_out/r05_tie_audit.txt 2>&1
tail -3 gpurun_out/r05_tie_audit.txt | cut -c1-400
grep "FLIPS [1-9]" gpurun_out/r05_tie_audit.txt | cut -c1-300
tools/abn.sh 3 ab/base.so ab/hdr.so -- --workload C5 2>&1 | tee gpurun_out/r05_hdr_c5_ab.txt
( time python -m pytest tests -m gpu -x -q ) > gpurun_out/r05_gputests_2.txt 2>&1; tail -5 gpurun_out/r05_gputests_2.txt
