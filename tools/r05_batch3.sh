#!/bin/bash
O=gpurun_out/r05_frame_small_launch.txt; : > $O
run() { echo "== $1" | tee -a $O; env $2 python tools/debug/frame_figure.py 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d['shipped_radius_0.5']
print('   r=2.0  gpu_ms_per_frame %.5f  direct %.5f  graph %s   |  r=0.5  gpu_ms_per_frame %.5f  direct %.5f  graph %s' % (d['gpu_ms_per_frame'], d['direct_ms_per_frame'], d['graph_ms_per_frame'], s['gpu_ms_per_frame'], s['direct_ms_per_frame'], s['graph_ms_per_frame']))" | tee -a $O; }
run "default (EASU TH=16, RCAS TH=16, serial below 4 images)" "X=1"
run "EASU TH=32 (RCAS 16)" "OVRFSR_EASU_TH=32"
run "RCAS TH=32 (EASU 16)" "OVRFSR_RCAS_TH=32"
run "both 32 (round 4 kernels), serial" "OVRFSR_EASU_TH=32 OVRFSR_RCAS_TH=32"
run "both 32, forked (round 4)" "OVRFSR_EASU_TH=32 OVRFSR_RCAS_TH=32 OVRFSR_SERIAL=0"
run "default again" "X=1"
tools/debug/frame_timeline.sh >> $O 2>&1
tail -22 $O
