"""k_long probe: a small deep-bin pool with every multi-read search sent to k_long (long_budget), with the round progress
on stderr (opts.debug, one line per host sync).  Run under `timeout`: a stuck kernel shows as progress lines that stop.
usage: long_probe.py n L G K budget [rounds_per_sync]"""
import sys, time
import spring_amd as sa

def main():
    n, L, G, K, b = (int(x) for x in sys.argv[1:6])
    rps = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    t0 = time.time()
    with sa.ReorderStage(sa.ReorderOpts(num_chains=K, num_thr=2, deep_bins=1, long_budget=b, debug=1, rounds_per_sync=rps)) as st:
        st.load_synth(n, L, G, 23, 10000)
        out = st.run().streams()
    print("done", n, L, G, K, b, "long_searches", out["stats"]["long_searches"], "%.2f s" % (time.time() - t0), flush=True)

if __name__ == "__main__":
    main()
