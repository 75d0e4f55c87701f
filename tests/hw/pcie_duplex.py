"""Is the host link full duplex here?  One-way and simultaneous two-way copy
rates between pinned host memory and the device (two HIP streams)."""
import time
import torch
n = 1 << 30
h1 = torch.empty(n, dtype=torch.uint8).pin_memory()
h2 = torch.empty(n, dtype=torch.uint8).pin_memory()
d1 = torch.empty(n, dtype=torch.uint8, device="cuda")
d2 = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(up, down, reps=4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        if up:
            with torch.cuda.stream(s1):
                d1.copy_(h1, non_blocking=True)
        if down:
            with torch.cuda.stream(s2):
                h2.copy_(d2, non_blocking=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


run(True, True, 1)
a, b, c = run(True, False), run(False, True), run(True, True)
print(f"H2D alone {n / a / 1e9:.1f} GB/s, D2H alone {n / b / 1e9:.1f} GB/s, "
      f"both at once {n / c / 1e9:.1f} GB/s each ({2 * n / c / 1e9:.1f} GB/s in all)")
