"""Is the GEMM limited by the memory system?  Same tiles / MFMA work, but all row tiles read the same 64 rows of A."""
import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/afford-motion_amd')
from afm import ops
dev = torch.device('cuda:0')
def t(fn, reps=100):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
M = 10432
for name, n, k in [("in_proj", 1536, 512), ("ffn2", 512, 1024)]:
    x = torch.randn(M, k, device=dev); w = torch.randn(n, k, device=dev) / k ** 0.5; b = torch.randn(n, device=dev)
    out = torch.empty(M, n, device=dev)
    d0 = t(lambda: ops.linear(x, w, b, out=out))
    d1 = t(lambda: ops.linear(x[:64].contiguous(), w, b, out=out, rows=M, a_map=(64, 0, 0)))
    small = torch.empty(64, n, device=dev)
    d2 = t(lambda: ops.linear(x, w, b, out=small, rows=M, c_map=(64, 0, 0)))
    d3 = t(lambda: ops.linear(x[:64].contiguous(), w, b, out=small, rows=M, a_map=(64, 0, 0), c_map=(64, 0, 0)))
    f = 2 * M * n * k / 1e12
    print(f"{name}: normal {f/d0:.1f} TF | A cache-resident {f/d1:.1f} | C collapsed to 64 rows {f/d2:.1f} | both {f/d3:.1f}")
