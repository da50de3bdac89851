"""Probe (GPU): cost of the pieces of a Gram-matrix BatchNorm for the frozen teacher's conv3 layers (1x1, P -> 4P at
M = 8 x 110 x 160): the current path (conv + tile statistics, finalize, apply with residual + ReLU) against
Gram matrix of the input (the weight-gradient kernel with x = dy) + the conv with a folded scale / shift + residual + ReLU epilogue."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openess_amd import hip, engine
import torch.nn as nn

def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

B, H, W = 8, 110, 160
tot_a = tot_b = 0.0
for P, nblk in ((64, 3), (128, 4), (256, 6), (512, 3)):
    C4 = 4 * P
    a2 = torch.randn(B, H, W, P, device="cuda").relu().bfloat16()
    res = torch.randn(B, H, W, C4, device="cuda").bfloat16()
    w = (torch.randn(C4, P, 1, 1, device="cuda") / P ** 0.5)
    bn = nn.BatchNorm2d(C4).cuda()
    packed = hip.pack_conv_weight(w)
    bias = torch.randn(C4, device="cuda")
    t_cur = timeit(lambda: hip.conv_bn_train_nhwc(a2, packed, C4, 1, 1, 1, 0, 1, bn, relu=True, residual=res))
    t_gram = timeit(lambda: hip.conv2d_wgrad(a2, a2, P, P, 1, 1, 1, 0, 1))
    t_fold = timeit(lambda: hip.conv2d_nhwc(a2, packed, bias, C4, 1, 1, 1, 0, 1, relu=True, residual=res))
    t_plain = timeit(lambda: hip.conv2d_nhwc(a2, packed, None, C4, 1, 1, 1, 0, 1))
    part = torch.empty(((B * H * W + 127) // 128, 2, C4), dtype=torch.float32, device="cuda")
    t_stats = timeit(lambda: hip.conv2d_nhwc(a2, packed, None, C4, 1, 1, 1, 0, 1, tile_stats=part))
    # correctness of the Gram statistics against the direct ones (fp64 on the host side of the probe)
    G = hip.conv2d_wgrad(a2, a2, P, P, 1, 1, 1, 0, 1).double().reshape(P, P)
    M = B * H * W
    mu = a2.double().reshape(M, P).mean(0)
    cov = G / M - torch.outer(mu, mu)
    wb = w.bfloat16().double().reshape(C4, P)
    var_g = ((wb @ cov) * wb).sum(1)
    y = a2.double().reshape(M, P) @ wb.t()
    var_d = y.var(0, unbiased=False)
    rel = ((var_g - var_d).abs() / var_d).max().item()
    print(f"P={P:4d} x{nblk}: current {t_cur:7.1f} us | gram {t_gram:6.1f} + folded conv(+res,relu) {t_fold:7.1f} = {t_gram + t_fold:7.1f} us"
          f" | plain conv {t_plain:6.1f}, conv+tile stats {t_stats:6.1f} | var rel err {rel:.2e}", flush=True)
    tot_a += nblk * t_cur; tot_b += nblk * (t_gram + t_fold)
print(f"teacher conv3+bn3 total: current {tot_a / 1e3:.2f} ms, gram form {tot_b / 1e3:.2f} ms (+ 16 small statistics launches)")
