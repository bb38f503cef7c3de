"""What this GPU's HBM delivers to simple streaming kernels, as context for the HBM-bound kernels' figures:
a read-only reduction, a copy and a fill over 4 GiB, timed with events.  python tools/hbm_probe.py"""
import torch


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = torch.device("cuda:0")
    n = 1 << 30                                   # 4 GiB of fp32
    x = torch.randn(n, device=dev)
    y = torch.empty_like(x)
    gb = n * 4 / 1e9
    ms = timed(lambda: x.sum())
    print(f"read  (torch.sum, 4 GiB):            {ms:.3f} ms  {gb / ms:.2f} TB/s")
    ms = timed(lambda: torch.amax(x))
    print(f"read  (torch.amax, 4 GiB):           {ms:.3f} ms  {gb / ms:.2f} TB/s")
    ms = timed(lambda: y.copy_(x))
    print(f"copy  (read 4 GiB + write 4 GiB):    {ms:.3f} ms  {2 * gb / ms:.2f} TB/s")
    ms = timed(lambda: y.fill_(1.0))
    print(f"write (fill, 4 GiB):                 {ms:.3f} ms  {gb / ms:.2f} TB/s")
    ms = timed(lambda: torch.add(x, 1.0, out=y))
    print(f"read + write (x + 1 -> y):           {ms:.3f} ms  {2 * gb / ms:.2f} TB/s")


if __name__ == "__main__":
    main()
