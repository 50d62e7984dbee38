"""Practical HBM roof: device-to-device copy bandwidth (read + write bytes / time) for a 16 GiB buffer."""
import torch
n = 16 << 30
a = torch.empty(n, dtype=torch.uint8, device="cuda")
b = torch.empty(n, dtype=torch.uint8, device="cuda")
a.fill_(1)
for _ in range(2):
    b.copy_(a)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
reps = 5
for _ in range(reps):
    b.copy_(a)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print("copy 16 GiB: %.2f ms -> %.2f TB/s (read + write)" % (ms, 2 * n / ms / 1e9))
e0.record()
for _ in range(reps):
    a.fill_(3)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print("fill 16 GiB: %.2f ms -> %.2f TB/s (write)" % (ms, n / ms / 1e9))
