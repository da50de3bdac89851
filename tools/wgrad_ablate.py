"""Time the weight-gradient kernel (+ its split-K reduce) on SemSegE2VID-decoder-like shapes (HIP events)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openess_amd import hip
SHAPES = {"r55": (8, 55, 80, 256, 256, 3), "u110": (8, 110, 160, 128, 128, 3), "u110b": (8, 110, 160, 256, 128, 3),
          "u220": (8, 220, 320, 64, 64, 3), "u220b": (8, 220, 320, 128, 64, 3), "u440": (8, 440, 640, 32, 32, 3),
          "u440b": (8, 440, 640, 64, 32, 3), "p440": (8, 440, 640, 32, 16, 1)}
N = int(os.environ.get("ABL_N", "10"))
for name in sys.argv[1:] or list(SHAPES):
    B, H, W, Cin, Cout, R = SHAPES[name]
    x = torch.randn(B, H, W, Cin, device="cuda").bfloat16()
    gy = torch.randn(B, H, W, Cout, device="cuda").bfloat16()
    for _ in range(3):
        hip.conv2d_wgrad(x, gy, Cout, Cin, R, R, 1, R // 2, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N):
        hip.conv2d_wgrad(x, gy, Cout, Cin, R, R, 1, R // 2, 1)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / N
    fl = 2.0 * B * H * W * Cout * Cin * R * R
    mb = (x.numel() + gy.numel()) * 2 / 1e6
    print(f"wgrad {name}: {ms*1e3:.1f} us  {fl / ms / 1e9:.0f} TF/s  inputs {mb:.0f} MB -> {mb / ms / 1e3:.2f} TB/s if read once", flush=True)
