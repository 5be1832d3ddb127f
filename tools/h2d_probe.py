#!/usr/bin/env python3
"""Raw host-to-device rate of page-locked buffers on this box (torch streams): what the H2D-inclusive figure can reach at most."""
import time, torch
torch.cuda.set_device(0)
for nstreams in (1, 2, 4, 8):
    for mb in (4, 20):
        n = mb << 20
        hs = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(nstreams)]
        ds = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(nstreams)]
        ss = [torch.cuda.Stream() for _ in range(nstreams)]
        def run(reps):
            for _ in range(reps):
                for h, d, s in zip(hs, ds, ss):
                    with torch.cuda.stream(s):
                        d.copy_(h, non_blocking=True)
            torch.cuda.synchronize()
        run(3)
        t0 = time.perf_counter(); run(20); dt = time.perf_counter() - t0
        print(f"{nstreams} streams x {mb} MB: {20 * nstreams * n / dt / 1e9:.1f} GB/s H2D")
# both directions at once
n = 20 << 20
h1 = torch.empty(n, dtype=torch.uint8).pin_memory(); d1 = torch.empty(n, dtype=torch.uint8, device="cuda")
h2 = torch.empty(n, dtype=torch.uint8).pin_memory(); d2 = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    with torch.cuda.stream(s1): d1.copy_(h1, non_blocking=True)
    with torch.cuda.stream(s2): h2.copy_(d2, non_blocking=True)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"H2D + D2H together: {20 * n / dt / 1e9:.1f} GB/s each way")
