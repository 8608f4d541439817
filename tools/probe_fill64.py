"""64x64-tile GEMM efficiency vs number of full rounds (1024 resident workgroup slots = 256 CUs x 4)."""
import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/afford-motion_amd')
from afm import ops
dev = torch.device('cuda:0')
def t(m, n, k, reps=50):
    x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev) / k ** 0.5; b = torch.randn(n, device=dev)
    out = torch.empty(m, n, device=dev)
    for _ in range(3): ops.linear(x, w, b, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): ops.linear(x, w, b, out=out)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for tiles in (512, 1024, 2048, 4096, 8192, 16384):
    m = tiles // 8 * 64
    for k in (512, 2048):
        dt = t(m, 512, k)
        print(f"tiles={tiles:6d} ({tiles/1024:5.2f} rounds) M={m:7d} N=512 K={k}: {dt*1e6:8.1f} us {2*m*512*k/dt/1e12:6.1f} TF/s")
