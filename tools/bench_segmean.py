"""K7 superpixel scatter-mean at the BASELINE size (B=8, 256 ch, 440x640): ms and algorithmic GB/s
(one read of the features + ids; fp32 features 2.31 GB, bf16 1.15 GB per call)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openess_amd import hip
B, C, H, W, sps = 8, 256, 440, 640, 100
yy = (torch.arange(H) * 10 // H)[:, None]; xx = (torch.arange(W) * 10 // W)[None, :]
ids_blocks = (yy * 10 + xx)[None].repeat(B, 1, 1).long().cuda()
ids_rand = torch.randint(0, 100, (B, H, W)).cuda()
ids_rand255 = torch.randint(0, 256, (B, H, W)).cuda()
# SAM-like maps: Voronoi cells of 180 random seeds per sample (irregular, spatially coherent regions, uint8 ids)
gy, gx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
vor = []
for b in range(B):
    g = torch.Generator().manual_seed(b)
    sy, sx = torch.rand(180, generator=g) * H, torch.rand(180, generator=g) * W
    vor.append(((gy[None] - sy[:, None, None]) ** 2 + (gx[None] - sx[:, None, None]) ** 2).argmin(0))
ids_vor = torch.stack(vor).long().cuda()
for dt in (torch.bfloat16, torch.float32):
    feat = torch.randn(B, H, W, C, device="cuda").to(dt).permute(0, 3, 1, 2)
    for name, ids in (("10x10 blocks", ids_blocks), ("voronoi cells (180 / map)", ids_vor), ("random per-pixel ids < 100", ids_rand),
                      ("random per-pixel ids < 256", ids_rand255)):
        for _ in range(3):
            hip.superpixel_pool(feat, ids, sps, S=(B - 1) * sps + 256)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            hip.superpixel_pool(feat, ids, sps, S=(B - 1) * sps + 256)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        gb = (feat.numel() * feat.element_size() + ids.numel() * 8) / 1e9
        print(f"segment mean {str(dt)[6:]:8s} {name:22s}: {ms:.3f} ms  {gb / ms * 1e3:.0f} GB/s algorithmic ({gb / ms * 1e3 / 80:.1f} % of 8 TB/s)")
