#!/usr/bin/env python3
"""A/B of the tuning options of spring_reorder_opts on one synthetic workload (results are identical by
construction; tests/test_gpu_parity.py::test_tuning_opts_do_not_change_results checks that).
usage: ab_search.py reads,readlen[,chains[,coverage[,err_ppm]]] name=k:v,k:v ...   (name=  -> defaults)
Prints one line per variant: chains-stage ms (best of 2), search-kernel avg launch us, rounds."""
import json
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spring_amd


def main():
    g = [int(x) for x in sys.argv[1].split(",")]
    n, L = g[0], g[1]
    K = g[2] if len(g) > 2 else 0
    cov = g[3] if len(g) > 3 else 25
    err = g[4] if len(g) > 4 else 10000
    G = max(n * L // cov, 2 * L)
    for spec in sys.argv[2:]:
        name, _, kv = spec.partition("=")
        kw = {k: (tuple(int(x) for x in v.split("/")) if "/" in v else int(v)) for k, v in (p.split(":") for p in kv.split(",") if p)}  # plan0:4/8/16
        best = None
        for it in range(3):
            with spring_amd.ReorderStage(spring_amd.ReorderOpts(device=0, num_chains=K, num_thr=8, time_search=(it == 2), **kw)) as s:
                s.load_synth(n, L, G, 11, err)
                s.run()
                st = s.stats()
            if it and (best is None or st["ms_chains"] < best["ms_chains"]) and it < 2:
                best = st
            if it == 2:
                tl = st
        print(json.dumps({"variant": name, "opts": kw, "ms_chains": round(best["ms_chains"], 1), "ms_dict": round(best["ms_dict"], 1),
                          "search_us": round(tl["ms_search_kernel"] * 1e3 / max(tl["search_launches"], 1), 1),
                          "ms_chains_timed": round(tl["ms_chains"], 1), "rounds": best["rounds"],
                          "singletons": best["n_single"], "unmatched": best["unmatched"]}), flush=True)


if __name__ == "__main__":
    main()
