#!/usr/bin/env python3
"""E2VID's three 5x5 stride-2 encoder ConvLayers at the BASELINE size (B = 8): time per launch, back to back (HIP events)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openess_amd import hip
for name,B,H,W,Cin,Cout in [("enc0",8,440,640,32,64),("enc1",8,220,320,64,128),("enc2",8,110,160,128,256)]:
    x=torch.randn(B,H,W,Cin,device='cuda').bfloat16(); w=torch.randn(Cout,Cin,5,5,device='cuda')*0.05
    pk=hip.pack_conv_weight(w); out=torch.empty(B,H//2,W//2,Cout,device='cuda',dtype=torch.bfloat16)
    for _ in range(3): hip.conv2d_nhwc(x,pk,None,Cout,5,5,2,2,1,out=out)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): hip.conv2d_nhwc(x,pk,None,Cout,5,5,2,2,1,out=out)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/20
    print(f"{os.environ.get('OESS_LIB_PATH','default'):32s} {name} {ms*1e3:7.1f} us {2.0*B*(H//2)*(W//2)*Cout*Cin*25/ms/1e9:7.1f} TF/s")
