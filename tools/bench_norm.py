"""BatchNorm(train)+ReLU forward / backward pieces on ResNet-50 activations: time and effective HBM rate (HIP events)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openess_amd import hip
SH = [(8, 220, 320, 64), (8, 110, 160, 64), (8, 110, 160, 256), (8, 55, 80, 128), (8, 55, 80, 512), (8, 55, 80, 256), (8, 55, 80, 1024), (8, 55, 80, 2048)]
N = 20
for (B, H, W, C) in SH:
    x = torch.randn(B, C, H, W, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    bn = torch.nn.BatchNorm2d(C).cuda().train()
    for mode in ("fwd", "fwd+bwd"):
        def run():
            y = hip.batch_norm_train(x, bn, relu=True)
            if mode == "fwd+bwd":
                y.backward(y.detach())
                x.grad = None
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(N): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / N
        mb = B * H * W * C * 2 / 1e6
        passes = 3 if mode == "fwd" else 3 + 5        # fwd: stats read + apply read/write; bwd: (y, dy) reduce + (y, dy) read + dx write
        print(f"{B}x{H}x{W}x{C} {mode}: {ms*1e3:.1f} us  tensor {mb:.0f} MB  ~{passes * mb / ms / 1e3:.2f} TB/s over {passes} passes", flush=True)
