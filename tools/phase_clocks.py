#!/usr/bin/env python3
"""Per-wavefront phase clocks of the round kernel (a library built with -DSR_PHASE_TIMING, selected with
SPRING_AMD_LIB; tools/r2_sweeps.sh builds it): the search half's time by outcome, in shader clocks, summed over
the run through the chains' statistics fields.  usage: phase_clocks.py reads readlen"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spring_amd

n, L = int(sys.argv[1]), int(sys.argv[2])
with spring_amd.ReorderStage(spring_amd.ReorderOpts(device=0, num_chains=0, num_thr=8)) as s:
    s.load_synth(n, L, n * L // 25, 11, 10000)
    s.run()
    st = s.stats()
n1, n2 = st["hits"], st["lost"] >> 32
n3 = st["n_matched"] - n1 - n2
print("chains stage %.1f ms, %d rounds (the clocks add a few %%)" % (st["ms_chains"], st["rounds"]))
print("search hit in the first ordered batch : %10d searches, %7.0f clocks each" % (n1, st["probes"] / max(n1, 1)))
print("search hit in the second ordered batch: %10d searches, %7.0f clocks each" % (n2, st["keyok"] / max(n2, 1)))
print("search hit in the tail (approx. count): %10d searches, %7.0f clocks each" % (n3, st["cands"] / max(n3, 1)))
print("failed searches (2 per unmatched read): %10d searches, %7.0f clocks each" % (2 * st["unmatched"], st["iterations"] / max(2 * st["unmatched"], 1)))
