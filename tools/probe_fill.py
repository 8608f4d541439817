"""GEMM efficiency at exactly-full grids (no tail): separates loop efficiency from wave quantisation."""
import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/afford-motion_amd')
from afm import ops
dev = torch.device('cuda:0')
def t(m, n, k, reps=100):
    x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev) / k ** 0.5; b = torch.randn(n, device=dev)
    out = torch.empty(m, n, device=dev)
    for _ in range(3): ops.linear(x, w, b, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): ops.linear(x, w, b, out=out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    return dt
for label, m, n, k in [("128x128 tiles, 252 WGs (1/CU)", 21 * 128, 1536, 512), ("128x128, 504 WGs (2/CU)", 42 * 128, 1536, 512),
                       ("128x128, 1008 WGs (2 rounds)", 84 * 128, 1536, 512), ("128x128, 2016 WGs (4 rounds)", 168 * 128, 1536, 512),
                       ("128x128 K=2048, 504 WGs", 42 * 128, 1536, 2048),
                       ("64x128 tiles (N=512), 512 WGs", 128 * 64, 512, 512), ("64x128 (N=512), 1024 WGs", 256 * 64, 512, 512),
                       ("64x128 K=1024, 512 WGs", 128 * 64, 512, 1024)]:
    dt = t(m, n, k)
    print(f"{label:34s} M={m:6d} N={n} K={k}: {dt*1e6:8.1f} us {2*m*n*k/dt/1e12:6.1f} TF/s")
