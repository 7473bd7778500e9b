"""Yardstick only (never on the product path): what the vendor GEMM library reaches on the small-K linear shapes of the
UNet's transformer blocks, HIP-event timed like tools/autotune.py — tells whether the fixed per-launch cost of our
kernels on these shapes is a hardware floor or ours."""
import torch

SHAPES = [(65536, 320, 320), (65536, 640, 320), (65536, 320, 1280), (16384, 640, 640), (16384, 1280, 640),
          (16384, 640, 2560), (4096, 1280, 1280), (4096, 2560, 1280), (4096, 1280, 5120), (65536, 2560, 320),
          (16384, 5120, 640), (4096, 10240, 1280), (1024, 1280, 1280)]


def main():
    dev = torch.device("cuda:0")
    print(f"{'M':>6} {'N':>6} {'K':>6} {'us':>8} {'TF/s':>7} {'GB/s':>7}")
    for M, N, K in SHAPES:
        a = torch.randn(M, K, device=dev, dtype=torch.float16)
        w = torch.randn(N, K, device=dev, dtype=torch.float16) * K ** -0.5
        for _ in range(3):
            torch.nn.functional.linear(a, w)
        best = 1e9
        for _ in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            torch.nn.functional.linear(a, w)
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3)
        # back-to-back x20: amortises the event/dispatch overhead
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            torch.nn.functional.linear(a, w)
        e1.record()
        e1.synchronize()
        b2b = e0.elapsed_time(e1) * 1e3 / 20
        print(f"{M:>6} {N:>6} {K:>6} {best:>8.1f} {2.0 * M * N * K / best / 1e6:>7.0f} "
              f"{(M * K + N * K + M * N) * 2 / best / 1e3:>7.0f}   back-to-back {b2b:>7.1f} us")


if __name__ == "__main__":
    main()
