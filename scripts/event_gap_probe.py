"""what an in-stream event record / wait costs between two dependent kernels (the executor records one event per weight
gradient on the launch stream): 2000 small kernels back to back, bare / with a record after each / with a record + a second
stream waiting on it / with a wait on an event the other stream recorded"""
import time
import torch
x = torch.zeros(1 << 20, device="cuda")
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
N = 2000


def run(mode):
    evs = [torch.cuda.Event(enable_timing=False) for _ in range(N)]
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(N):
        x.add_(1.0)
        if mode >= 1:
            evs[i].record(main)
        if mode >= 2:
            side.wait_event(evs[i])
        if mode >= 3:
            with torch.cuda.stream(side):
                x[:1024].add_(0.0) if False else None
    torch.cuda.synchronize()
    return (time.time() - t0) / N * 1e6


for rep in range(2):
    for mode, name in ((0, "bare"), (1, "record after each kernel"), (2, "record + other stream waits")):
        print("%-32s %.2f us per kernel" % (name, run(mode)))
# events with timing enabled (what the profiling pass uses)
evs = [torch.cuda.Event(enable_timing=True) for _ in range(N)]
torch.cuda.synchronize()
t0 = time.time()
for i in range(N):
    x.add_(1.0)
    evs[i].record(main)
torch.cuda.synchronize()
print("%-32s %.2f us per kernel" % ("record (timing enabled)", (time.time() - t0) / N * 1e6))
# device-side view: one event pair around the whole loop
for mode, name in ((0, "bare"), (1, "record after each kernel"), (2, "record + other stream waits")):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    evs = [torch.cuda.Event(enable_timing=False) for _ in range(N)]
    torch.cuda.synchronize()
    e0.record()
    for i in range(N):
        x.add_(1.0)
        if mode >= 1:
            evs[i].record(main)
        if mode >= 2:
            side.wait_event(evs[i])
    e1.record()
    torch.cuda.synchronize()
    print("device time %-28s %.2f us per kernel" % (name, e0.elapsed_time(e1) / N * 1e3))
