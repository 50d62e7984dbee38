"""CPU baseline port scaling on this host: orc_reorder_omp at several thread counts (same 8 M-read sample)."""
import sys, time, os
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import spring_amd
from oracle import pyoracle as po
n, L = 8_000_000, 150
dna = spring_amd.synth_dna_host(n, L, n * L // 25, 18, 10000)
read, ln = po.load_dna(dna, n, L)
print("host cpus:", os.cpu_count())
for T in (32, 64, 128, 192, 256):
    t0 = time.perf_counter(); po.reorder_omp(read, ln, L, T); el = time.perf_counter() - t0
    print("T=%3d  %.2f s  %.3f Mreads/s" % (T, el, n / el / 1e6), flush=True)
