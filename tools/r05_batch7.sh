#!/bin/bash
export OVRFSR_LIB=$PWD/ab/audit.so
O=gpurun_out/r05_audit_search.txt; : > $O
AUDIT=1 timeout 900 python tools/debug/easu_err_search.py 300 1 0,1,2,3 2>&1 | grep -v "champion patch" | tee -a $O
AUDIT=1 timeout 900 python tools/debug/easu_err_search.py 300 7 0,2 2>&1 | grep -v "champion patch" | tee -a $O
HALF=1 HSCALE=1 AUDIT=1 timeout 600 python tools/debug/easu_err_search.py 200 2 0,2 2>&1 | tee -a $O
HALF=1 HSCALE=40 AUDIT=1 timeout 600 python tools/debug/easu_err_search.py 200 3 0,2 2>&1 | tee -a $O
