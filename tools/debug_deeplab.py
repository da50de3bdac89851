import sys, json
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from oracle import nets as on, losses as ol
from tests.synth import fill_by_name
from openess_amd.models.deeplabv3 import deeplabv3_resnet50
from openess_amd import hip
keys = json.load(open('tests/golden/nets_keys.json'))
def cos(a, b):
    a = a.float().cpu().flatten().double(); b = b.flatten().double(); return float(a @ b / (a.norm() * b.norm() + 1e-30))
for (H, W) in ((64, 96), (192, 256)):
    torch.manual_seed(1)
    net = deeplabv3_resnet50(11, None, 32, ''); fill_by_name(net, 15); net.cuda().train(); net.classifier.ASPP.project[3].p = 0.0
    ref = on.DeepLabV3(11, 32); fill_by_name(ref, 15, keys['deeplab']); ref.train(); ref.classifier.ASPP.project[3].p = 0.0
    img = torch.rand(2, 3, H, W); tgt = torch.randint(0, 11, (2, H, W))
    lg, _ = net(img.cuda()); loss, _ = hip.task_loss(lg, tgt.cuda(), 11); loss.backward()
    lr, _ = ref(img); lossr = ol.task_loss(lr, tgt, 11); lossr.backward()
    print(H, W, 'loss', float(loss), float(lossr))
    pr = dict(ref.named_parameters())
    for n, p in net.named_parameters():
        if p.grad is None or n not in pr or pr[n].grad is None: continue
        if n.endswith('weight') and p.ndim == 4 and ('classifier' in n or n in ('backbone.conv1.weight', 'backbone.layer1.0.conv1.weight', 'backbone.layer4.2.conv3.weight', 'backbone.layer3.0.conv2.weight')):
            print('   ', n, round(cos(p.grad, pr[n].grad), 4))
