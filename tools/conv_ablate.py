"""Time one conv shape (HIP events).  A/B two builds of the library on the same box with OESS_LIB_PATH=<other liboess.so>."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openess_amd import hip
SHAPES = {"gates": (8, 110, 160, 256, 512, 3, 1, 1, 1), "l3": (8, 55, 80, 256, 256, 3, 1, 4, 4),
          "pw": (8, 55, 80, 256, 1024, 1, 1, 0, 1), "l4": (8, 55, 80, 512, 512, 3, 1, 8, 8),
          "head": (8, 440, 640, 8, 32, 5, 1, 2, 1), "enc3": (8, 110, 160, 128, 256, 5, 2, 2, 1), "enc2": (8, 220, 320, 64, 128, 5, 2, 2, 1),
          "res": (8, 55, 80, 256, 256, 3, 1, 1, 1), "aspp": (8, 28, 40, 2048, 256, 3, 1, 6, 6), "dl4": (8, 28, 40, 512, 512, 3, 1, 2, 2), "gk4": (8, 110, 160, 1024, 512, 3, 1, 1, 1), "gk1": (8, 110, 160, 64, 512, 3, 1, 1, 1),
          "g1": (8, 220, 320, 128, 256, 3, 1, 1, 1), "g3": (8, 55, 80, 512, 1024, 3, 1, 1, 1),
          "enc1": (8, 440, 640, 32, 64, 5, 2, 2, 1),
          "t1": (8, 110, 160, 512, 2048, 1, 1, 0, 1), "t2": (8, 110, 160, 256, 1024, 1, 1, 0, 1), "t3": (8, 110, 160, 1024, 256, 1, 1, 0, 1), "t4": (8, 110, 160, 128, 512, 1, 1, 0, 1), "t5": (8, 110, 160, 2048, 512, 1, 1, 0, 1), "t6": (8, 110, 160, 1024, 2048, 1, 1, 0, 1), "k64": (8, 220, 320, 64, 256, 1, 1, 0, 1), "k64c128": (8, 220, 320, 64, 128, 1, 1, 0, 1), "k64c64": (8, 220, 320, 64, 64, 1, 1, 0, 1), "k64c512": (8, 220, 320, 64, 512, 1, 1, 0, 1), "k128": (8, 220, 320, 128, 256, 1, 1, 0, 1), "k256": (8, 220, 320, 256, 256, 1, 1, 0, 1), "k512": (8, 220, 320, 512, 256, 1, 1, 0, 1)}
for name in sys.argv[1:]:
    B, H, W, Cin, Cout, R, st, pad, dil = SHAPES[name]
    mode = os.environ.get("ABL_DATA", "randn")
    x = torch.randn(B, H, W, Cin, device="cuda").bfloat16()
    w = torch.randn(Cout, Cin, R, R, device="cuda") * 0.05
    if mode == "zeros": x.zero_(); w.zero_()
    if mode == "ones": x.fill_(1.0); w.fill_(1.0)
    pk = hip.pack_conv_weight(w)
    Ho, Wo = (H + 2 * pad - dil * (R - 1) - 1) // st + 1, (W + 2 * pad - dil * (R - 1) - 1) // st + 1
    out = torch.empty(B, Ho, Wo, Cout, device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        hip.conv2d_nhwc(x, pk, None, Cout, R, R, st, pad, dil, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(int(os.environ.get('ABL_N', '20'))):
        hip.conv2d_nhwc(x, pk, None, Cout, R, R, st, pad, dil, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / int(os.environ.get('ABL_N', '20'))
    fl = 2.0 * B * Ho * Wo * Cout * Cin * R * R
    print(f"{name}: {ms:.4f} ms  {fl / ms / 1e9:.0f} TF/s", flush=True)
