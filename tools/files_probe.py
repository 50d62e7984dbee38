#!/usr/bin/env python3
"""The drop-in call on a temp directory with the library's phase clocks on stderr (opts.debug): where
`stage_incl_files` spends its time.  usage: files_probe.py reads readlen [repeats] [tmp_root]"""
import os
import shutil
import sys
import tempfile
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spring_amd

n, L = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
root = sys.argv[4] if len(sys.argv) > 4 else None
G = n * L // 25
with spring_amd.ReorderStage(spring_amd.ReorderOpts(device=0, num_thr=8)) as s:
    s.load_synth(n, L, G, 11)
    host = np.frombuffer(s.download_dna(), np.uint8).copy()
for it in range(reps):
    td = tempfile.mkdtemp(prefix="spring_files_", dir=root)
    with open(os.path.join(td, "input_clean_1.dna"), "wb") as f:
        f.write(host.tobytes())
    os.sync()
    t0 = time.perf_counter()
    spring_amd.call_reorder(td, spring_amd.CompressionParams(L, [n, 0], num_thr=8),
                            spring_amd.ReorderOpts(device=0, num_thr=8, debug=True))
    t = time.perf_counter() - t0
    ob = sum(os.path.getsize(os.path.join(td, f)) for f in os.listdir(td))
    shutil.rmtree(td, ignore_errors=True)
    print("run %d: %.3f s = %.1f Mreads/s  (in %.2f GB, out %.2f GB, dir %s)" % (it, t, n / t / 1e6, host.nbytes / 1e9, ob / 1e9, td), flush=True)
