import torch, time
dev = torch.device('cuda:0')
for M in (65536, 1 << 20):
    for (n, k) in ((1024, 1024), (1024, 512), (1, 1024)):
        dz = torch.randn(M, n, device=dev, dtype=torch.bfloat16)
        y = torch.randn(M, k, device=dev, dtype=torch.bfloat16)
        S = 32
        a = dz.view(S, M // S, n).transpose(1, 2); b = y.view(S, M // S, k)
        forms = {'one GEMM bf16       ': lambda: (dz.t() @ y).float(),
                 'bmm bf16 + sum(f32)  ': lambda: torch.bmm(a, b).sum(0, dtype=torch.float32),
                 'bmm out_dtype f32+sum': lambda: torch.bmm(a, b, out_dtype=torch.float32).sum(0)}
        for name, f in forms.items():
            for _ in range(3): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter(); e0.record()
            for _ in range(10): f()
            e1.record(); host = (time.perf_counter() - t0) / 10 * 1e3
            torch.cuda.synchronize()
            print(f'M={M:8d} n={n:4d} k={k:4d} {name} gpu {e0.elapsed_time(e1)/10:7.3f} ms   host issue {host:7.3f} ms')
