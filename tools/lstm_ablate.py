"""Time the fused ConvLSTM kernel on the three E2VID recurrent layers (HIP events); A/B two builds with OESS_LIB_PATH."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openess_amd import hip
# B, H, W, C_in(x), C_hidden
SHAPES = {"l1": (8, 220, 320, 64, 64), "l2": (8, 110, 160, 128, 128), "l3": (8, 55, 80, 256, 256)}
N = int(os.environ.get("ABL_N", "20"))
for name in sys.argv[1:] or list(SHAPES):
    B, H, W, Cx, C = SHAPES[name]
    xh = torch.randn(B, H, W, Cx + C, device="cuda").bfloat16()
    w = torch.randn(4 * C, Cx + C, 3, 3, device="cuda") * 0.02
    bias = torch.zeros(4 * C, device="cuda")
    pk = hip.pack_conv_weight(w, flip=2)
    cell = torch.zeros(B, H, W, C, device="cuda")
    hid = torch.empty(B, H, W, C, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        hip.convlstm_fused(xh, pk, bias, cell, hid, 3, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N):
        hip.convlstm_fused(xh, pk, bias, cell, hid, 3, 1)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / N
    fl = 2.0 * B * H * W * 4 * C * (Cx + C) * 9
    print(f"lstm {name}: {ms:.4f} ms  {fl / ms / 1e9:.0f} TF/s", flush=True)
