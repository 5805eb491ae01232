"""Which BLAS backend serves the SST token GEMMs ([V,128] x [128,256], bf16) faster on this box."""
import time, torch
dev = torch.device('cuda')
x = torch.randn(259761, 128, device=dev, dtype=torch.bfloat16)
w = torch.randn(256, 128, device=dev, dtype=torch.bfloat16)
b = torch.randn(256, device=dev, dtype=torch.bfloat16)
for lib in ('cublaslt', 'cublas'):
    torch.backends.cuda.preferred_blas_library(lib)
    for name, fn in (('linear+bias', lambda: torch.nn.functional.linear(x, w, b)), ('matmul', lambda: x @ w.t()),
                     ('addmm', lambda: torch.addmm(b, x, w.t()))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        print(lib, name, round((time.perf_counter() - t0) / 20 * 1e6, 1), 'us')
