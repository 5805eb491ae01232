"""Library bf16 GEMM forms of the decoder's backward at 1 M rows: dX = dz W and dW = dz^T y, operand layouts."""
import torch, time
dev = torch.device('cuda:0')
M = 1 << 20
def t(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (n, k) in ((1024, 1024), (1024, 512)):
    dz = torch.randn(M, n, device=dev, dtype=torch.bfloat16)
    y = torch.randn(M, k, device=dev, dtype=torch.bfloat16)
    W = torch.randn(n, k, device=dev, dtype=torch.bfloat16)          # nn.Linear weight [out, in]
    Wt = W.t().contiguous()                                           # [in, out]
    fl = 2.0 * M * n * k
    r = {}
    r['dX = dz @ W            '] = t(lambda: dz @ W)
    r['dX = dz @ Wt.t()       '] = t(lambda: dz @ Wt.t())
    r['dX = linear(dz, Wt)    '] = t(lambda: torch.nn.functional.linear(dz, Wt))
    r['dW = dz.t() @ y        '] = t(lambda: dz.t() @ y)
    r['dW = (y.t() @ dz).t()  '] = t(lambda: (y.t() @ dz).t())
    dzt = dz.t().contiguous()
    r['dW = dzt @ y (dzt cont)'] = t(lambda: dzt @ y)
    print(f'n={n} k={k}')
    for kk, v in r.items():
        print(f'  {kk} {v:7.3f} ms  {fl / v / 1e9:7.1f} TFLOP/s')

print('sliced dW (batched GEMM over row slices, summed):')
for (n, k) in ((1024, 1024), (1024, 512)):
    dz = torch.randn(M, n, device=dev, dtype=torch.bfloat16)
    y = torch.randn(M, k, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * M * n * k
    ref = (dz[:65536].float().t() @ y[:65536].float())
    for S in (8, 16, 32, 64):
        a = dz.view(S, M // S, n).transpose(1, 2)
        b = y.view(S, M // S, k)
        try:
            f = lambda: torch.bmm(a, b, out_dtype=torch.float32).sum(0)
            ms = t(f)
            print(f'  n={n} k={k} S={S:3d} bmm(out_dtype=f32).sum  {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s')
        except Exception as e:
            print('  out_dtype not available:', type(e).__name__, str(e)[:100])
            f = lambda: torch.bmm(a, b).float().sum(0)
            ms = t(f)
            print(f'  n={n} k={k} S={S:3d} bmm(bf16).float().sum   {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s')
