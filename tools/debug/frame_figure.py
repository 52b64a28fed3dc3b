# tools/debug/frame_figure.py -- the `frame` object of the bench line alone (the reference's debug-mode figure, one stereo pair per frame)
import json, sys; sys.path.insert(0, '.')
import argparse, torch
import bench
args = bench.parse_args(["--pairs", "1", "--workload", "C2"] + sys.argv[1:])
s = bench.GpuShard(0, 0, args); s.bind()
for _ in range(20): s.step()
s.sync()
print(json.dumps(bench.frame_leg(args, s)))
