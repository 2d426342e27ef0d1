"""Host cost of torch's stream primitives on this machine (microseconds per call)."""
import time

import torch

dev = torch.device("cuda", 0)
main = torch.cuda.current_stream(dev)
side = torch.cuda.Stream(device=dev)
x = torch.zeros(16, device=dev)
torch.cuda.synchronize()


def bench(name, fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"{name:44s} {(t1 - t0) / n * 1e6:8.2f} us", flush=True)


def ctx():
    with torch.cuda.stream(side):
        pass


def ev():
    e = torch.cuda.Event()
    e.record(main)
    return e


e0 = ev()
bench("with torch.cuda.stream(side): pass", ctx)
bench("side.wait_stream(main)", lambda: side.wait_stream(main))
bench("Event() + record", ev)
bench("side.wait_event(e)", lambda: side.wait_event(e0))
bench("torch.cuda.current_stream(dev)", lambda: torch.cuda.current_stream(dev))
bench("current_stream(dev).cuda_stream", lambda: torch.cuda.current_stream(dev).cuda_stream)
bench("torch.empty(1024, uint8)", lambda: torch.empty(1024, dtype=torch.uint8, device=dev))
bench("x.add_(1) (one small kernel)", lambda: x.add_(1))
bench("torch.cuda.set_stream(side); set_stream(main)", lambda: (torch.cuda.set_stream(side), torch.cuda.set_stream(main)))

# device-side cost of handing work from one stream to another: a chain of tiny kernels, all on one stream /
# alternating between two streams with an event between every two
n = 200
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        x.add_(1)
    torch.cuda.synchronize()
    one = (time.perf_counter() - t0) / n * 1e6
    t0 = time.perf_counter()
    for i in range(n):
        s, o = (side, main) if i % 2 else (main, side)
        s.wait_stream(o)
        with torch.cuda.stream(s):
            x.add_(1)
    torch.cuda.synchronize()
    two = (time.perf_counter() - t0) / n * 1e6
    print(f"chain of small kernels: one stream {one:.2f} us per kernel, alternating streams {two:.2f} us per kernel", flush=True)
