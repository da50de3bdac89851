"""Full-size timing of the MaskCLIP ViT-B/16 tower (B=8, 440x640 -> 28x40 patches + cls = 1121 tokens)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openess_amd.models.maskclip_model import maskClipFeatureExtractor
torch.manual_seed(0)
m = maskClipFeatureExtractor(text_categories=11).cuda().eval()
img = torch.rand(8, 3, 440, 640, device="cuda")
for _ in range(3):
    out = m(img)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    out = m(img)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
L, C, B = 1121, 768, 8
gemm = 2.0 * B * L * (11 * (C * 3 * C + C * C) + (C * C + C * C) + 12 * 2 * C * 4 * C) + 2.0 * B * 1120 * (3 * 256 * C + C * 512 + 512 * 11)
attn = 11 * 2 * 2.0 * B * 12 * L * L * 64
print(f"maskclip tower B=8 440x640: {ms:.2f} ms  ({(gemm + attn) / ms / 1e9:.0f} TFLOP/s over {(gemm + attn) / 1e9:.0f} GFLOP; out {tuple(out.shape)})")
