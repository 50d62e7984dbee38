import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, spring_amd
n, L = 100_000_000, 150
with spring_amd.ReorderStage(spring_amd.ReorderOpts(num_chains=0, num_thr=8)) as st:
    st.load_synth(n, L, n * L // 25, 11)
    st.run()
    s = st.streams()
f = s["flag"] == ord("0")
starts = np.flatnonzero(f)
sizes = np.diff(np.append(starts, len(f)))
print("contigs", len(sizes), "mean", sizes.mean(), "max", sizes.max())
for t in (64, 128, 254, 1000):
    print("reads beyond the %d-th of their contig: %.2f %%" % (t, 100.0 * np.maximum(sizes - t, 0).sum() / sizes.sum()))
print("quantiles", np.quantile(sizes, [0.5, 0.9, 0.99, 0.999]))
