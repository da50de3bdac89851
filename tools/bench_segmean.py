"""K7 superpixel scatter-mean at the BASELINE size (B=8, 256 ch, 440x640): ms and algorithmic GB/s
(one read of the features + ids; fp32 features 2.31 GB, bf16 1.15 GB per call)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openess_amd import hip
B, C, H, W, sps = 8, 256, 440, 640, 100
yy = (torch.arange(H) * 10 // H)[:, None]; xx = (torch.arange(W) * 10 // W)[None, :]
ids_blocks = (yy * 10 + xx)[None].repeat(B, 1, 1).long().cuda()
ids_rand = torch.randint(0, 100, (B, H, W)).cuda()
for dt in (torch.bfloat16, torch.float32):
    feat = torch.randn(B, H, W, C, device="cuda").to(dt).permute(0, 3, 1, 2)
    for name, ids in (("10x10 blocks", ids_blocks), ("random per-pixel ids", ids_rand)):
        for _ in range(3):
            hip.superpixel_pool(feat, ids, sps, S=B * sps)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            hip.superpixel_pool(feat, ids, sps, S=B * sps)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        gb = (feat.numel() * feat.element_size() + ids.numel() * 8) / 1e9
        print(f"segment mean {str(dt)[6:]:8s} {name:22s}: {ms:.3f} ms  {gb / ms * 1e3:.0f} GB/s algorithmic ({gb / ms * 1e3 / 80:.1f} % of 8 TB/s)")
