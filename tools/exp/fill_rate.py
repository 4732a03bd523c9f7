import sys, time, torch
sys.path.insert(0, "/root/repo")
from legionsolvers_amd import Context, _lib
from legionsolvers_amd.backend import _ptr
ctx = Context(torch.device("cuda", 0)); be = ctx.backend
n = 503_000_000
a = torch.empty(n, dtype=torch.float64, device="cuda"); b = torch.empty(n, dtype=torch.float64, device="cuda")
def timed(fn, reps=5):
    fn(); ctx.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        ctx.synchronize(); best = min(best, (time.perf_counter() - t0) / reps * 1e3)
    return best
for nt in (-1, 0):
    _lib.call("lsk_set_option", b"blas1.nontemporal", nt)
    t = timed(lambda: _lib.call("lsk_fill_f64", n, 1.5, _ptr(a), be.stream()))
    print(f"nt={nt} fill 4.02 GB: {t:.3f} ms {n*8/t/1e9:.2f} TB/s")
    t = timed(lambda: _lib.call("lsk_copy_f64", n, _ptr(a), _ptr(b), be.stream()))
    print(f"nt={nt} copy 4.02 GB -> 4.02 GB: {t:.3f} ms {2*n*8/t/1e9:.2f} TB/s")
    t = timed(lambda: _lib.call("lsk_dot_host_f64", n, _ptr(a), _ptr(a), _ptr(b), be.stream())) if False else 0
