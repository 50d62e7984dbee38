#!/usr/bin/env python3
"""Host <-> device legs of the stage on their own: load_dna from a pageable host buffer, and the download of every
output stream.  usage: pcie_probe.py reads readlen"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spring_amd

n, L = int(sys.argv[1]), int(sys.argv[2])
G = n * L // 25
with spring_amd.ReorderStage(spring_amd.ReorderOpts(device=0, num_thr=8)) as s:
    s.load_synth(n, L, G, 11)
    host = np.frombuffer(s.download_dna(), np.uint8).copy()
for it in range(2):
    with spring_amd.ReorderStage(spring_amd.ReorderOpts(device=0, num_thr=8)) as s:
        t0 = time.perf_counter()
        s.load_dna(host, n, L)
        t1 = time.perf_counter()
        s.run()
        t2 = time.perf_counter()
        res = s.streams()
        t3 = time.perf_counter()
    nb = sum(v.nbytes for v in res.values() if hasattr(v, "nbytes"))
    print("load_dna %.3f s (%.1f GB/s)  stage %.3f s  streams %.3f s (%.1f GB/s)  total %.3f s = %.1f Mreads/s" % (
        t1 - t0, host.nbytes / (t1 - t0) / 1e9, t2 - t1, t3 - t2, nb / (t3 - t2) / 1e9, t3 - t0, n / (t3 - t0) / 1e6), flush=True)
