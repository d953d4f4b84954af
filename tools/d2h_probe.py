"""Raw device -> host copy rate of the box (pinned and pageable, one stream and two), next to what tools/pcie_inclusive.py sees through
volume_renderer(empty_gpu_cache=True): is the [4096, 512] alpha hand-over bound by the link or by the renderer?"""
import time
import torch

dev = "cuda"
for mb in (8, 64, 512):
    n = mb * (1 << 20) // 4
    src = torch.rand(n, device=dev)
    dst = torch.empty(n, pin_memory=True)
    for _ in range(2):
        dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    reps = max(4, 2048 // mb)
    t0 = time.perf_counter()
    for _ in range(reps):
        dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    one = (time.perf_counter() - t0) / reps
    s2 = torch.cuda.Stream()
    half = n // 2
    t0 = time.perf_counter()
    for _ in range(reps):
        dst[:half].copy_(src[:half], non_blocking=True)
        with torch.cuda.stream(s2):
            dst[half:].copy_(src[half:], non_blocking=True)
    torch.cuda.synchronize()
    two = (time.perf_counter() - t0) / reps
    page = torch.empty(n)
    t0 = time.perf_counter()
    page.copy_(src)
    torch.cuda.synchronize()
    pg = time.perf_counter() - t0
    print(f"{mb:4d} MiB  pinned, one stream {mb / 1024 / one:6.2f} GiB/s   two streams {mb / 1024 / two:6.2f} GiB/s   pageable {mb / 1024 / pg:6.2f} GiB/s")
