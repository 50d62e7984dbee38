"""tools/overlap_probe2.py -- tools/overlap_probe.py for any pool: ONE stage over a pool against TWO independent stages over
half-size pools of the same kind side by side (each on its own stream and host thread), library defaults: what two chain
groups could buy on deep-coverage / contended / genome-like pools.

  python tools/overlap_probe2.py n,L,G[,flags]  ...      (G = genome length; flags: gen = genome-like)
"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spring_amd  # noqa: E402


def prepared(n, L, G, seed, err, K=0):
    s = spring_amd.ReorderStage(spring_amd.ReorderOpts(num_chains=K, num_thr=8, phases=1))
    s.load_synth(n, L, G, seed, err)
    s.build_dict()
    return s


def chains_wall(stages):
    bar = threading.Barrier(len(stages) + 1)
    def work(s):
        bar.wait()
        s.run_chains()
    th = [threading.Thread(target=work, args=(s,)) for s in stages]
    for t in th:
        t.start()
    bar.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    return time.perf_counter() - t0


for a in sys.argv[1:]:
    f = a.split(",")
    n, L, G = int(f[0]), int(f[1]), int(f[2])
    err = 10000 | (0x20000000 if len(f) > 3 and f[3] == "gen" else 0)
    for rep in range(2):
        s = prepared(n, L, G, 21, err)
        w1 = chains_wall([s])
        st = s.stats()
        K = st["chains"]
        s.close()
        ss = [prepared(n // 2, L, max(G // 2, 4 * L) if G > 100000 else G, 21 + i, err, K // 2) for i in range(2)]
        w2 = chains_wall(ss)
        k2 = ss[0].stats()["chains"]
        for x in ss:
            x.close()
        print("%s: one stage (K=%d, alts=%d, %d rounds) chains %.1f ms | two half pools side by side (K=%d each) %.1f ms  (%.2fx)" % (
            a, K, st["alternatives"], st["rounds"], w1 * 1e3, k2, w2 * 1e3, w1 / w2), flush=True)
