#!/bin/bash
export OVRFSR_LIB=$PWD/ab/audit.so
O=gpurun_out/r05_audit_search_long.txt; : > $O
AUDIT=1 timeout 500 python tools/debug/easu_err_search.py 1500 11 0,1,2,3 2>&1 | grep -v "champion patch\|generation *\(1\|10\|50\|100\|200\|500\|1000\):" | tee -a $O
AUDIT=1 timeout 300 python tools/debug/easu_err_search.py 1500 23 2,3 2>&1 | grep -v "champion patch\|generation *\(1\|10\|50\|100\|200\|500\|1000\):" | tee -a $O
HALF=1 HSCALE=6 AUDIT=1 timeout 240 python tools/debug/easu_err_search.py 800 5 0,2 2>&1 | grep -v "generation *\(1\|10\|50\|100\|200\|500\):" | tee -a $O
HALF=1 HSCALE=40 AUDIT=1 timeout 240 python tools/debug/easu_err_search.py 800 6 1,3 2>&1 | grep -v "generation *\(1\|10\|50\|100\|200\|500\):" | tee -a $O
OVRFSR_LIB=$PWD/ab/audit.so timeout 300 python tools/debug/tie_audit.py 2.0 2>&1 | tail -1 | tee -a $O
