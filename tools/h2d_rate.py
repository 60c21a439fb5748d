import torch, time
x = torch.empty(64, 1, 768, 1024, dtype=torch.float32).pin_memory()
d = torch.empty_like(x, device="cuda")
for _ in range(3): d.copy_(x, non_blocking=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): d.copy_(x, non_blocking=True)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("H2D 64 x 768x1024 fp32 (%.1f MB) pinned: %.3f ms per step -> %.1f GB/s, %.1f us per image" % (x.numel() * 4 / 1e6, ms, x.numel() * 4 / ms / 1e6, ms * 1e3 / 64))
