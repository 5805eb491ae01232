"""dW = dY^T X for tall-skinny operands (rows = points of a batch, 10^5; 24..144 features): one library GEMM against
the same contraction split over the rows into a batched GEMM + a sum.  Run on the GPU box."""
import torch
dev = torch.device('cuda:0')
for n, cin, cout in ((131072, 131, 32), (131072, 144, 32), (131072, 128, 144), (131072, 32, 144), (131072, 16, 3), (1048576, 512, 1024)):
    x = torch.randn(n, cin, device=dev)
    gy = torch.randn(n, cout, device=dev)

    def plain():
        return gy.t() @ x

    def split(r=2048):
        s = n // r
        out = (gy[:s * r].view(s, r, cout).transpose(1, 2) @ x[:s * r].view(s, r, cin)).sum(0)
        if s * r < n:
            out = out + gy[s * r:].t() @ x[s * r:]
        return out

    for name, f in (('plain', plain), ('split 2048', split), ('split 4096', lambda: split(4096)), ('split 1024', lambda: split(1024))):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        print(n, cin, cout, name, round(e0.elapsed_time(e1) / 10 * 1e3, 1), 'us', 'rel err vs plain', float((f() - plain()).norm() / plain().norm()))
