"""Achievable HBM streaming rates on this GPU for the access mixes of the HBM-bound kernels (read-only, write-only,
copy), measured with plain torch kernels on 1 GiB fp32 tensors - the practical ceilings the K1 / K7 figures should be
read against (the 8 TB/s datasheet number is not reachable by a write-dominated kernel)."""
import torch
n = 1 << 28                       # 1 GiB of fp32
x = torch.empty(n, device="cuda"); y = torch.empty(n, device="cuda")
def t(fn, it=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
gb = n * 4 / 1e9
ms = t(lambda: x.zero_());            print(f"write only (fill 1 GiB)     : {ms:.3f} ms  {gb / ms * 1e3:.0f} GB/s")
ms = t(lambda: x.fill_(1.5));         print(f"write only (fill_ value)    : {ms:.3f} ms  {gb / ms * 1e3:.0f} GB/s")
ms = t(lambda: x.sum());              print(f"read only (sum 1 GiB)       : {ms:.3f} ms  {gb / ms * 1e3:.0f} GB/s")
ms = t(lambda: y.copy_(x));           print(f"copy (1 GiB read + 1 GiB wr): {ms:.3f} ms  {2 * gb / ms * 1e3:.0f} GB/s")
ms = t(lambda: torch.add(x, 1.0, out=y)); print(f"read + write elementwise   : {ms:.3f} ms  {2 * gb / ms * 1e3:.0f} GB/s")
