import sys, json, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from oracle import nets as on
from tests.synth import fill_by_name
from openess_amd.models.image_model import DilationFeatureExtractor
from openess_amd import engine
keys = json.load(open('tests/golden/nets_keys.json'))
g = dict(np.load('tests/golden/nets.npz'))
t = DilationFeatureExtractor(None); fill_by_name(t.encoder, 13); fill_by_name(t.decoder[0], 14); t.cuda().train()
ref = on.DilationFeatureExtractor(); fill_by_name(ref.encoder, 13, keys['teacher_encoder']); fill_by_name(ref.decoder[0], 14); ref.train()
img = torch.from_numpy(g['teacher_img'])
def rel(a, b):
    a = a.float().cpu(); return float((a - b).abs().max() / b.abs().max()), float(torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0))
with torch.no_grad():
    x = engine.to_cl_bf16(img.cuda()); e = t.encoder
    c1 = e.conv1(x); r1 = ref.encoder.conv1(img); print('conv1', rel(c1, r1))
    x = engine.batch_norm_act(c1, e.bn1, relu=True); rr = torch.relu(ref.encoder.bn1(r1)); print('bn1', rel(x, rr))
    x = e.maxpool(x); rr = ref.encoder.maxpool(rr); print('pool', rel(x, rr))
    for ln in ('layer1', 'layer2', 'layer3', 'layer4'):
        for bi, (blk, rblk) in enumerate(zip(getattr(e, ln), getattr(ref.encoder, ln))):
            x = blk(x); rr = rblk(rr); print(ln, bi, rel(x, rr))
