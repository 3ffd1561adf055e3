#!/usr/bin/env python3
"""Does hipStreamWaitEvent hold a second stream back until the first stream's kernel has finished?  (Measurement tool; PyTorch
only as a convenient way to issue the HIP calls.)  A long kernel on s1 writes a buffer, an event without timing is recorded
behind it, s2 waits for the event and copies the buffer to page-locked host memory; repeated with the event object reused."""
import torch
dev = torch.device("cuda:0")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
n = 1 << 24
a = torch.zeros(n, device=dev)
h = torch.empty(n).pin_memory()
ev = torch.cuda.Event(enable_timing=False)
bad = 0
for it in range(1, 41):
    with torch.cuda.stream(s1):
        for _ in range(20):  # ~ms of work
            a.add_(1.0)
        ev.record(s1)
    with torch.cuda.stream(s2):
        s2.wait_event(ev)
        h.copy_(a, non_blocking=True)
        done = torch.cuda.Event(enable_timing=False)
        done.record(s2)
    done.synchronize()
    want = 20.0 * it
    if not bool((h == want).all()):
        bad += 1
print("iterations with a stale copy:", bad, "of 40")
