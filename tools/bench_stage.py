#!/usr/bin/env python3
"""One hot-path stage in a loop, for `rocprofv3 --kernel-trace --stats`:
    python tools/bench_stage.py deeplab_fwd|teacher_fwd|maskclip_fwd|deeplab_step [--iters N]
Full BASELINE size (B=8, 440x640), train-mode BatchNorm, random-init weights."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("stage")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--breakdown", action="store_true", help="per-shape HIP-event timing of the MFMA conv launches")
    a = ap.parse_args()
    dev = torch.device("cuda")
    torch.manual_seed(0)
    img = torch.rand(a.B, 3, 440, 640, device=dev)
    if a.stage == "deeplab_fwd" or a.stage == "deeplab_step":
        from openess_amd.models.deeplabv3 import deeplabv3_resnet50
        m = deeplabv3_resnet50(num_classes=11, text_embeddings_path='', output_stride=32, pretrained_backbone='').to(dev).train()
        gflop = 106.8 * a.B * (3 if a.stage == "deeplab_step" else 1)
    elif a.stage == "teacher_fwd":
        from openess_amd.models.image_model import DilationFeatureExtractor
        m = DilationFeatureExtractor(None).to(dev).train()
        gflop = 845.1 * a.B
    elif a.stage == "maskclip_fwd":
        from openess_amd.models.maskclip_model import maskClipFeatureExtractor
        m = maskClipFeatureExtractor(text_categories=11).to(dev).eval()
        gflop = 1859.7 * a.B / 8
    else:
        raise SystemExit(a.stage)

    def run():
        if a.stage == "deeplab_step":
            from openess_amd import hip
            for p in m.parameters():
                p.grad = None
            lg, _ = m(img)
            loss, _ = hip.task_loss(lg, tgt, 11)
            loss.backward()
        else:
            with torch.no_grad():
                m(img)
    tgt = torch.randint(0, 11, (a.B, 440, 640), device=dev)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    if a.breakdown:
        from openess_amd import hip
        hip.conv_timing_begin()
        run()
        st = hip.conv_timing_end()
        tot = 0.0
        for k, (n, tm, fl) in sorted(st["by_shape"].items(), key=lambda kv: -kv[1][1]):
            tot += tm
            print(f"# conv HxWxCin->Cout k,s,d {k}: {n:3d}x {tm / n * 1e3:8.1f} us each {tm:7.3f} ms {fl / tm / 1e9:7.1f} TF/s")
        print(f"# conv (Cout > 64) total {tot:.3f} ms, {st['flops'] / tot / 1e9:.1f} TF/s over {st['flops'] / 1e9:.0f} GFLOP")
    print(f"{a.stage}: {ms:.3f} ms  {gflop / ms:.1f} TFLOP/s over {gflop:.0f} GFLOP ({gflop / ms / 25:.1f} % of 2.5 PF)")


if __name__ == "__main__":
    main()
