import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import spring_amd
n, L, G, K = 10000000, 100, 5400, 8192
with spring_amd.ReorderStage(spring_amd.ReorderOpts(num_chains=K, num_thr=8, collect_stats=True)) as s:
    s.load_synth(n, L, G, 21, 10000); s.run(); st = s.stats()
it = st["iterations"]
print({k: st[k] for k in ("iterations","probes","keyok","cands","hits","lost","rounds","unmatched","n_single")})
print("per iteration: probes %.1f keyok %.1f cands %.1f ; lost/iter %.2f ; chain-rounds %d vs iterations %d" % (st["probes"]/it, st["keyok"]/it, st["cands"]/it, st["lost"]/it, st["rounds"]*K, it))
