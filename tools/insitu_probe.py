import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openess_amd import hip
B,H,W,Cin,Cout=8,28,40,1024,256
x=torch.randn(B,H,W,Cin,device='cuda').bfloat16()
w=torch.randn(Cout,Cin,1,1,device='cuda')*0.05
pk=hip.pack_conv_weight(w)
M=B*H*W
part=torch.empty(((M+63)//64,2,Cout),device='cuda')
out=torch.empty(B,H,W,Cout,device='cuda',dtype=torch.bfloat16)
def timeit(fn,n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ev=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a,b in ev:
        pre()
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts=sorted(a.elapsed_time(b)*1e3 for a,b in ev)
    return ts[len(ts)//2], ts[0]
pre=lambda: None
print('conv no stats        ', timeit(lambda: hip.conv2d_nhwc(x,pk,None,Cout,1,1,1,0,1,out=out)))
print('conv stats           ', timeit(lambda: hip.conv2d_nhwc(x,pk,None,Cout,1,1,1,0,1,out=out,tile_stats=part)))
y=torch.empty_like(x)
pre=lambda: torch.add(x,1,out=y)
print('producer then conv(y)', timeit(lambda: hip.conv2d_nhwc(y,pk,None,Cout,1,1,1,0,1,out=out,tile_stats=part)))
big=torch.empty(64<<20,device='cuda')
pre=lambda: big.fill_(1.0)
print('256MB fill then conv ', timeit(lambda: hip.conv2d_nhwc(x,pk,None,Cout,1,1,1,0,1,out=out,tile_stats=part)))
pre=lambda: None
# back-to-back enqueue without host gaps: 20 convs in a row, total
torch.cuda.synchronize()
a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50): hip.conv2d_nhwc(x,pk,None,Cout,1,1,1,0,1,out=out,tile_stats=part)
b.record(); torch.cuda.synchronize()
print('50 back-to-back, us each', a.elapsed_time(b)*1e3/50)
import time
t0=time.perf_counter()
for _ in range(200): hip.conv2d_nhwc(x,pk,None,Cout,1,1,1,0,1,out=out,tile_stats=part)
t1=time.perf_counter(); torch.cuda.synchronize()
print('host enqueue us per conv call', (t1-t0)/200*1e6)
