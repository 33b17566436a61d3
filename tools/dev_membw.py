"""Developer aid: what the chip sustains for plain fills / copies of the sizes the convolution epilogues write (torch kernels,
HIP-graph replays, in-stream events)."""
import torch
for mb in (26, 85, 340):
    n = mb * 1024 * 1024 // 2
    xs = [torch.empty(n, dtype=torch.bfloat16, device='cuda') for _ in range(4)]
    ys = [torch.randn(n, device='cuda').to(torch.bfloat16) for _ in range(4)]
    for name, fn, byt in (("fill", lambda i: xs[i].zero_(), 2 * n), ("copy", lambda i: xs[i].copy_(ys[i]), 4 * n), ("relu", lambda i: torch.relu_(ys[i]), 4 * n)):
        for i in range(4): fn(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(5):
                for i in range(4): fn(i)
        g.replay(); torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); e.record(); torch.cuda.synchronize()
        us = a.elapsed_time(e) * 1e3 / 20
        print(f"{name} {mb:4d} MB: {us:7.2f} us  {byt / us / 1e6:6.2f} TB/s", flush=True)
