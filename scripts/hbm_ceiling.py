"""dev helper: practical HBM ceilings on this box with plain torch ops (copy = read + write, fill = write only,
sum = read only), to put the per-kernel GB/s of DESIGN.md next to what the memory system delivers."""
import time, torch
dev = torch.device("cuda:0")
n = 1 << 30      # 4 GiB fp32 (larger than the 256 MiB Infinity Cache)
a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
b = torch.empty_like(a)


def t(fn, bytes_, name, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name:12s} {bytes_ / dt / 1e12:6.2f} TB/s  ({dt * 1e3:.2f} ms)", flush=True)


t(lambda: b.copy_(a), 8 * n, "copy r+w")
t(lambda: b.fill_(1.0), 4 * n, "fill w")
t(lambda: a.sum(), 4 * n, "sum r")
t(lambda: torch.add(a, 1.0, out=b), 8 * n, "add r+w")
