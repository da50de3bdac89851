"""What does the vendor GEMM (hipBLASLt via torch.mm) reach on the conv-equivalent GEMM shapes with random data?
A practical ceiling for the implicit-GEMM kernel (same M, N, K; no im2col gather, no epilogue fusion)."""
import torch
shapes = {"lstm1": (563200, 256, 1152), "lstm2": (140800, 512, 2304), "lstm3": (35200, 1024, 4608),
          "pw 256->1024": (140800, 1024, 256), "t1 512->2048": (140800, 2048, 512), "t3 1024->256": (140800, 256, 1024),
          "t4 128->512": (140800, 512, 128), "t5 2048->512": (140800, 512, 2048), "t6 1024->2048": (140800, 2048, 1024),
          "big square": (8192, 8192, 8192)}
import sys
if len(sys.argv) > 1:
    shapes = {k: v for k, v in shapes.items() if any(k.startswith(a) for a in sys.argv[1:])}
for name, (M, N, K) in shapes.items():
    for mode in ("randn", "zeros"):
        a = torch.randn(M, K, device="cuda").bfloat16()
        b = torch.randn(N, K, device="cuda").bfloat16()
        if mode == "zeros":
            a.zero_(); b.zero_()
        for _ in range(5):
            c = a @ b.t()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            c = a @ b.t()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 50
        print(f"{name:14s} {mode:6s} M={M} N={N} K={K}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.0f} TF/s", flush=True)
