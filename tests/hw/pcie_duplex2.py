"""Opposite-direction copies issued a few ms apart on two streams: does the
later (H2D) one wait for the earlier (D2H) one?"""
import time
import torch
n = 1 << 30
m = 547 * 10**6
h1 = torch.empty(n, dtype=torch.uint8).pin_memory()
h2 = torch.empty(n, dtype=torch.uint8).pin_memory()
d1 = torch.empty(n, dtype=torch.uint8, device="cuda")
d2 = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for trial in range(3):
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    ea = torch.cuda.Event(enable_timing=True)
    eb = torch.cuda.Event(enable_timing=True)
    e0.record()
    t0 = time.perf_counter()
    with torch.cuda.stream(s2):
        h2.copy_(d2, non_blocking=True)          # D2H 1 GiB
        ea.record(s2)
    time.sleep(0.003)
    t1 = time.perf_counter()
    with torch.cuda.stream(s1):
        d1[:m].copy_(h1[:m], non_blocking=True)  # H2D 0.55 GB
        eb.record(s1)
    s1.synchronize()
    t2 = time.perf_counter()
    s2.synchronize()
    t3 = time.perf_counter()
    print(f"trial {trial}: H2D issued at {1e3*(t1-t0):.1f} ms, done at {1e3*(t2-t0):.1f} ms; "
          f"D2H done at {1e3*(t3-t0):.1f} ms")
