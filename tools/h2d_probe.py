#!/usr/bin/env python3
"""Host -> device time of one bench step's frames (8 clips x 10 frames x 3 x 800 x 800), fp32 and uint8, pageable and pinned (GPU box)."""
import time, torch
dev = torch.device("cuda:0")
for name, t in (("fp32 [80,3,800,800]", torch.rand(80, 3, 800, 800)), ("uint8 [80,800,800,3]", (torch.rand(80, 800, 800, 3) * 255).to(torch.uint8))):
    for pin in (False, True):
        h = t.pin_memory() if pin else t
        d = torch.empty_like(t, device=dev)
        for _ in range(2):
            d.copy_(h, non_blocking=pin); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            d.copy_(h, non_blocking=pin)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        mb = t.numel() * t.element_size() / 1e6
        print(f"{name:22s} {'pinned' if pin else 'pageable':8s} {mb:7.1f} MB  {ms:7.2f} ms  {mb / ms:6.1f} GB/s")
