"""f1 measurement: FASTQ text (fixed-width records built with numpy) -> load_fastq; reports device time and GB/s of text."""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
import spring_amd
n, L = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(3)
rec = 10 + L + 3 + L + 1  # "@rDDDDDDD\n" + read + "\n+\n" + quality + "\n"
a = np.empty((n, rec), np.uint8)
a[:, 0] = ord("@"); a[:, 1] = ord("r")
idx = np.arange(n)
for d in range(7):
    a[:, 8 - d] = ord("0") + (idx // 10 ** d) % 10
a[:, 9] = ord("\n")
a[:, 10:10 + L] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (n, L))]
a[rng.random(n) < 0.02, 10 + L // 2] = ord("N")
a[:, 10 + L] = ord("\n"); a[:, 11 + L] = ord("+"); a[:, 12 + L] = ord("\n")
a[:, 13 + L:13 + 2 * L] = ord("I")
a[:, 13 + 2 * L] = ord("\n")
text = a.tobytes()
for it in range(2):
    t0 = time.perf_counter()
    with spring_amd.ReorderStage() as s:
        info = s.load_fastq(text)
    wall = time.perf_counter() - t0
print("n=%d L=%d text=%.2f GB  device=%.1f ms (%.0f GB/s of text, %.0f Mreads/s)  wall incl. H2D=%.3f s  clean=%d N=%d" % (
    n, L, len(text) / 1e9, info["ms_device"], len(text) / info["ms_device"] / 1e6, n / info["ms_device"] / 1e3, wall,
    info["num_reads_clean"][0], info["num_reads_N"][0]))
