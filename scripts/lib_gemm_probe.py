"""How fast does the library GEMM (hipBLASLt through torch.matmul) run the GEMM shapes of ResNet-50's 3x3 layers?
A yardstick for the implicit-GEMM kernels only: nothing in the package calls it."""
import torch, time
dev = "cuda"
for (M, N, K) in [(78400, 256, 2304), (313600, 128, 1152), (19600, 512, 4608), (1254400, 64, 576), (78400, 1024, 256),
                  (8192, 8192, 8192)]:
    a = torch.randn(M, K, device=dev, dtype=torch.float16)
    b = torch.randn(N, K, device=dev, dtype=torch.float16)
    for _ in range(3):
        c = a @ b.t()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        c = a @ b.t()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"M {M:8d} N {N:5d} K {K:5d}: {dt * 1e6:8.1f} us  {2.0 * M * N * K / dt / 1e12:7.1f} TF", flush=True)
