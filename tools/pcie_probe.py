#!/usr/bin/env python3
"""Measurement tool (not product code): what the GPU box's PCIe link gives page-locked copies — host -> device, device ->
host, both at once on two streams — for the batch sizes of the streaming leg of bench.py (12.7 MB up, 9.9 MB down per
4096-frame batch).  Uses PyTorch only as a convenient way to issue hipMemcpyAsync from Python."""
import json
import time

import torch

dev = torch.device("cuda:0")
out = {}
for mb in (1, 10, 13, 64):
    n = mb * 1024 * 1024
    h_up = torch.empty(n, dtype=torch.uint8).pin_memory()
    h_dn = torch.empty(n, dtype=torch.uint8).pin_memory()
    d_up = torch.empty(n, dtype=torch.uint8, device=dev)
    d_dn = torch.empty(n, dtype=torch.uint8, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    reps = 50

    def run(up, dn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            if up:
                with torch.cuda.stream(s1):
                    d_up.copy_(h_up, non_blocking=True)
            if dn:
                with torch.cuda.stream(s2):
                    h_dn.copy_(d_dn, non_blocking=True)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    run(True, True)
    t_up, t_dn, t_both = run(True, False), run(False, True), run(True, True)
    out[f"{mb}MB"] = {"h2d_GBps": n * reps / t_up / 1e9, "d2h_GBps": n * reps / t_dn / 1e9,
                      "both_each_GBps": n * reps / t_both / 1e9, "both_total_GBps": 2 * n * reps / t_both / 1e9}
print(json.dumps(out, indent=1))
