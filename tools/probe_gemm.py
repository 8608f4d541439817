"""Sustained timing of the encoder GEMM shapes (measurement tooling)."""
import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/afford-motion_amd')
from afm import ops, ffi
dev = torch.device('cuda:0')
M = 32 * 326
shapes = [("in_proj", M, 1536, 512), ("ffn1", M, 1024, 512), ("out_proj", M, 512, 512), ("ffn2", M, 512, 1024),
          ("motion_adapter", 32 * 196, 512, 263), ("motion_layer", 32 * 196, 263, 512)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for name, m, n, k in shapes:
    x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev) / k ** 0.5; b = torch.randn(n, device=dev)
    out = torch.empty(m, n, device=dev)
    for _ in range(3): ops.linear(x, w, b, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): ops.linear(x, w, b, out=out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f"{name:15s} M={m} N={n} K={k}: {dt*1e6:8.1f} us  {2*m*n*k/dt/1e12:6.1f} TF/s")
