import sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
from spring_amd import order_ops as oo
n, nN = 100_000_000, 3_000_000
rng = np.random.default_rng(1)
order = rng.permutation(n).astype(np.uint32)
order_N = np.sort(rng.choice(n + nN, nN, replace=False)).astype(np.uint32)
for name, f in (("generate_order_se", lambda: oo.generate_order_se(order)), ("generate_order_pe", lambda: oo.generate_order_pe(order)),
                ("correct_order", lambda: oo.correct_order(order, order_N, n))):
    f(); _, ms = f()
    print("%s n=%d kernel_ms=%.3f  (%.1f G entries/s)" % (name, n, ms, n / ms / 1e6))
