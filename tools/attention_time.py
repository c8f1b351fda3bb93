"""Time of the detection head's cross attention (200 queries x 32400 keys, 8 heads of 16, B = 2) through isf_attention_forward
and its error against a float64 softmax attention on a few queries.   gpurun -- python tools/attention_time.py"""
import torch, time, sys
sys.path.insert(0, '.')
import isfusion_amd
from isfusion_amd import fusion_ops as ops
dev = torch.device('cuda', 0)
B, P, HW, E = 2, 200, 32400, 128
g = torch.Generator().manual_seed(0)
q = torch.randn((B * P, E), generator=g).to(dev)
kv = torch.randn((B * HW, 2 * E), generator=g).to(dev)
ref = None
for _ in range(3):
    out = ops.attention(q, kv, kv[:, E:], B, P, HW, E, 8, ldkv=2 * E)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    out = ops.attention(q, kv, kv[:, E:], B, P, HW, E, 8, ldkv=2 * E)
e1.record(); torch.cuda.synchronize()
# fp64 reference on a few queries
qd = q.double().view(B, P, 8, 16)[:, :8]; kd = kv[:, :E].double().view(B, HW, 8, 16); vd = kv[:, E:].double().view(B, HW, 8, 16)
s = torch.einsum('bqhd,bkhd->bhqk', qd, kd) / 4.0
o = torch.einsum('bhqk,bkhd->bqhd', torch.softmax(s, -1), vd).reshape(B, 8, E)
err = (out.view(B, P, E)[:, :8].double() - o).abs().max().item()
print("head cross attention 200 x 32400: %.1f us per call, max err vs fp64 %.2e" % (e0.elapsed_time(e1) * 20, err))
