"""Do the branches of a captured HIP graph run concurrently on this ROCm?  Two spin kernels (torch.cuda._sleep) on
forked streams: wall time of a replay = one sleep if they overlap, two if the graph serialises its branches."""
import os
import sys
import time

os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')
import torch

dev = torch.device('cuda:0')
cyc = int(sys.argv[1]) if len(sys.argv) > 1 else 200000


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def serial():
    torch.cuda._sleep(cyc)
    torch.cuda._sleep(cyc)


def forked():
    cur = torch.cuda.current_stream()
    s2.wait_stream(cur)
    with torch.cuda.stream(s2):
        torch.cuda._sleep(cyc)
    torch.cuda._sleep(cyc)
    cur.wait_stream(s2)


print('eager serial  us', round(timeit(serial), 1))
print('eager forked  us', round(timeit(forked), 1))
for name, fn in (('serial', serial), ('forked', forked)):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s1):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s1):
            fn()
    torch.cuda.synchronize()
    print(f'graph {name:7s} us', round(timeit(g.replay), 1))
