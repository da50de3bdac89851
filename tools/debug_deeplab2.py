import sys, json
sys.path.insert(0, '/root/repo')
import numpy as np, torch, torch.nn.functional as F
from oracle import nets as on, losses as ol
from tests.synth import fill_by_name
from openess_amd.models import deeplabv3 as D, _resnet as R
from openess_amd import hip, engine
keys = json.load(open('tests/golden/nets_keys.json'))
def cos(a, b):
    a = a.detach().float().cpu().flatten().double(); b = b.detach().flatten().double(); return float(a @ b / (a.norm() * b.norm() + 1e-30))
H, W = 64, 96
def run(mode):
    torch.manual_seed(1)
    net = D.deeplabv3_resnet50(11, None, 32, ''); fill_by_name(net, 15); net.cuda().train(); net.classifier.ASPP.project[3].p = 0.0
    ref = on.DeepLabV3(11, 32); fill_by_name(ref, 15, keys['deeplab']); ref.train(); ref.classifier.ASPP.project[3].p = 0.0
    img = torch.rand(2, 3, H, W); tgt = torch.randint(0, 11, (2, H, W))
    lg, _ = net(img.cuda())
    if mode == 'torchloss':
        loss = ol.task_loss(lg.float().cpu(), tgt, 11) if False else torch.nn.functional.cross_entropy(lg.float(), tgt.cuda())
        lr, _ = ref(img); lossr = torch.nn.functional.cross_entropy(lr, tgt)
    else:
        loss, _ = hip.task_loss(lg, tgt.cuda(), 11)
        lr, _ = ref(img); lossr = ol.task_loss(lr, tgt, 11)
    loss.backward(); lossr.backward()
    pr = dict(ref.named_parameters())
    print(mode, float(loss.detach()), float(lossr.detach()))
    for n in ('classifier.text_embeddings', 'classifier.classifier.1.weight', 'classifier.classifier.0.weight', 'classifier.ASPP.project.0.weight'):
        p = dict(net.named_parameters())[n]
        if n in pr and pr[n].grad is not None: print('   ', n, round(cos(p.grad, pr[n].grad), 4))
run('hiploss')
run('torchloss')
orig = R.HipConv2d.forward
def torch_fwd(self, x):
    return F.conv2d(x.float()[:, :self.weight.shape[1]], self.weight, self.bias, self.stride, self.padding, self.dilation).to(torch.bfloat16)
R.HipConv2d.forward = torch_fwd
run('torchconv_hiploss')
