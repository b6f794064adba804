import sys, torch
sys.path.insert(0, '/root/repo')
from walkgpt_amd import autograd as ag, train_head
from walkgpt_amd.utils_walkgpt import SegAwareGate
dev = torch.device('cuda:0')
def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))
g = torch.Generator().manual_seed(1)
gate = SegAwareGate(1024)
with torch.no_grad():
    for k, p in gate.named_parameters():
        p.copy_((torch.randn(p.shape, generator=g) * (0.3 if p.dim() == 1 else p.shape[-1] ** -0.5)).to(torch.bfloat16).float())
    gate.net[0].weight.add_(1.0)
ref = SegAwareGate(1024); ref.load_state_dict(gate.state_dict())
gate = gate.to(dev).bfloat16()
x = torch.randn(2, 300, 1024, generator=g).to(torch.bfloat16)
dy = torch.randn(2, 300, 1024, generator=g).to(torch.bfloat16)
xh = x.to(dev).requires_grad_(True)
y = train_head._gate(gate, xh); y.backward(dy.to(dev))
xr = x.float().requires_grad_(True)
yr = xr * torch.sigmoid(ref.net(xr)); yr.backward(dy.float())
print("y", rel(y, yr), "dx", rel(xh.grad, xr.grad))
for (k, p), (_, q) in zip(gate.named_parameters(), ref.named_parameters()):
    print(k, rel(p.grad, q.grad), float(p.grad.float().norm()), float(q.grad.norm()))
# pieces
l = torch.randn(2, 300, 1, generator=g)
lh = l.to(dev).requires_grad_(True); xh2 = x.to(dev).requires_grad_(True)
y2 = ag.sigmoid_gate(xh2, lh); y2.backward(dy.to(dev))
lr = l.clone().requires_grad_(True); xr2 = x.float().requires_grad_(True)
(xr2 * torch.sigmoid(lr)).backward(dy.float())
print("gate op: dx", rel(xh2.grad, xr2.grad), "dl", rel(lh.grad, lr.grad))
