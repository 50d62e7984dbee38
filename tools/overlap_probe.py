"""tools/overlap_probe.py -- does the GPU fill one round kernel's drain with another stream's round kernel?

A round costs ~46 us + 158 us per 65 536 chains: the fixed part is the drain of k_round_mc (a wavefront of
four chains lives 50-60 us, the last of 3.2 generations leaves the chip four fifths empty), k_mg_mark and two kernel
boundaries.  This probe runs TWO independent stages (each half the reads, half the chains, its own stream, its own host
thread) side by side and compares the wall clock of their chain stages with ONE stage of the whole size: the upper bound of
what a schedule with two chain groups in anti-phase could buy.

  python tools/overlap_probe.py [n_total] [K_total]
"""
import ctypes as C
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import spring_amd  # noqa: E402
from spring_amd import _lib  # noqa: E402

L_ = _lib.lib()
L = 150


def make_pool(n, seed):
    G = n * L // 25
    nb = L_.spring_synth_dna_bytes(n, L)
    buf = torch.empty(nb, dtype=torch.uint8, device="cuda")
    assert L_.spring_synth_dna_device(C.c_void_p(buf.data_ptr()), n, L, G, seed, 10000) == 0
    torch.cuda.synchronize()
    return buf, nb


def prepared(n, K, seed):
    buf, nb = make_pool(n, seed)
    s = spring_amd.ReorderStage(spring_amd.ReorderOpts(num_chains=K, num_thr=8, fused=3))
    s.load_dna_device(buf.data_ptr(), nb, n, L, True)
    s.build_dict()
    return s, buf


def chains_wall(stages):
    """run_chains of every stage in its own thread; -> wall seconds from the common start to the last one's end"""
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


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    for rep in range(2):
        s, b = prepared(n, K, 11)
        w = chains_wall([s])
        st = s.stats()
        print("one stage   n=%d K=%d: chains wall %.1f ms (stage clock %.1f ms, %d rounds)" % (n, K, w * 1e3, st["ms_chains"], st["rounds"]), flush=True)
        s.close(); del b
        s, b = prepared(n // 2, K // 2, 11)
        w = chains_wall([s])
        st = s.stats()
        print("half alone  n=%d K=%d: chains wall %.1f ms (stage clock %.1f ms, %d rounds)" % (n // 2, K // 2, w * 1e3, st["ms_chains"], st["rounds"]), flush=True)
        s.close(); del b
        s1, b1 = prepared(n // 2, K // 2, 11)
        s2, b2 = prepared(n // 2, K // 2, 12)
        w = chains_wall([s1, s2])
        print("two halves  2 x (n=%d K=%d) side by side: chains wall %.1f ms (stage clocks %.1f / %.1f ms, %d / %d rounds)" % (
            n // 2, K // 2, w * 1e3, s1.stats()["ms_chains"], s2.stats()["ms_chains"], s1.stats()["rounds"], s2.stats()["rounds"]), flush=True)
        s1.close(); s2.close(); del b1, b2
        if rep == 0:  # four quarters as well
            ss = [prepared(n // 4, K // 4, 11 + i) for i in range(4)]
            w = chains_wall([x[0] for x in ss])
            print("four quarters 4 x (n=%d K=%d) side by side: chains wall %.1f ms" % (n // 4, K // 4, w * 1e3), flush=True)
            for x in ss:
                x[0].close()
            del ss


if __name__ == "__main__":
    main()
