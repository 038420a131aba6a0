#!/usr/bin/env python3
"""Aggregate images/s of the CPU restatement for several (processes x threads) shapes on this host (no GPU involved): what bench.cpu_baseline's
worker processes should look like.  usage: cpu_workers_probe.py 8x32 16x16 8x16 ..."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = '''
import sys, os, json
sys.path.insert(0, %r)
sys.argv = ["bench.py", "--no-cpu-baseline"]
import bench
class A: pass
a = A(); a.cpu_seconds = 10.0; a.height = 480; a.width = 640
w = bench.wait_cpu_workers(bench.spawn_cpu_workers(a))
print(json.dumps(bench.collect_cpu_workers(w, 10.0)))
''' % ROOT
for shape in sys.argv[1:] or ["8x32"]:
    n, t = shape.split("x")
    env = dict(os.environ, CASAPOSE_CPU_WORKERS=n, CASAPOSE_CPU_WORKER_THREADS=t)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(shape, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
