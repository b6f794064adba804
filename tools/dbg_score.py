import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from walkgpt_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (lh, img, inp, orig) in [(128, 512, (384, 512), (75, 111)), (256, 1024, (1024, 683), (448, 299)), (256, 1024, (1024, 1024), (448, 448))]:
    m = (torch.randn(3, 1, lh, lh) * 4).to(dev)
    a = ops.postprocess_masks(m, img, inp, orig)[:, 0]
    b, _ = ops.postprocess_masks_scored(m, img, inp, orig)
    d = (a - b).abs()
    bad = (d > 0).nonzero()
    print(orig, "mismatches", bad.shape[0], "max", d.max().item(), "first", bad[:5].tolist(), "rows", sorted(set(bad[:, 1].tolist()))[:20])
