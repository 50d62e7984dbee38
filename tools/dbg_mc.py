import sys, os, time, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import spring_amd
n, L, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
G = max(n * L // 25, 4 * L)
with spring_amd.ReorderStage(spring_amd.ReorderOpts(device=0, num_chains=K, num_thr=8, rounds_per_sync=int(sys.argv[4]) if len(sys.argv) > 4 else 0)) as s:
    s.load_synth(n, L, G, 11, 10000)
    t = time.time(); s.run(); st = s.stats()
    print("ok n=%d K=%d rounds=%d single=%d %.3fs" % (n, K, st["rounds"], st["n_single"], time.time() - t), flush=True)
