import torch, time
dev = torch.device('cuda:0')
M = 262144
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (n, k) in ((128, 128), (256, 128), (128, 256), (384, 128)):
    dz = torch.randn(M, n, device=dev, dtype=torch.bfloat16)
    y = torch.randn(M, k, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * M * n * k
    print(f'n={n} k={k}: one GEMM {t(lambda: dz.t() @ y)*1e3:7.1f} us', end='')
    for S in (16, 64, 128, 256):
        a = dz.view(S, M // S, n).transpose(1, 2); b = y.view(S, M // S, k)
        ms = t(lambda: torch.bmm(a, b).sum(0, dtype=torch.float32))
        print(f' | S={S}: {ms*1e3:6.1f} us ({fl/ms/1e9:5.0f} TF/s)', end='')
    print()
