"""Is the fp32 gradient residue conditioning or a bug?  Compare torch-CPU fp32 and the HIP fp32 path against
a torch-CPU fp64 run of the oracle on the same batch."""
import copy, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flair-1_amd"))
import torch, torch.nn as nn
import flair_amd
from oracle import unet_resnet34 as om

H, W = int(sys.argv[1]) if len(sys.argv) > 1 else 160, int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda:0")
ref = om.seeded_model(5, 13, 123)
hip = flair_amd.create_model("unet", "resnet34", encoder_weights=None, in_channels=5, classes=13, compute_dtype="f32")
hip.load_state_dict(ref.state_dict()); hip = hip.to(dev).train()
ref64 = copy.deepcopy(ref).double().train(); ref.train()
g = torch.Generator().manual_seed(5)
x = torch.randn(2, 5, H, W, generator=g); lab = torch.randint(0, 13, (2, H, W), generator=g)
l64 = ref64(x.double()); nn.functional.cross_entropy(l64, lab).backward()
l32 = ref(x); nn.functional.cross_entropy(l32, lab).backward()
lh = hip(x.to(dev)); nn.functional.cross_entropy(lh, lab.to(dev)).backward()
print("logits err vs fp64: cpu32 %.3e hip %.3e" % ((l32.double() - l64).abs().max(), (lh.cpu().double() - l64).abs().max()))
p64 = dict(ref64.named_parameters()); p32 = dict(ref.named_parameters())
rows = []
for k, p in hip.named_parameters():
    g64 = p64[k].grad; den = g64.abs().max() + 1e-12
    rows.append((k, float((p32[k].grad.double() - g64).abs().max() / den), float((p.grad.cpu().double() - g64).abs().max() / den)))
rows.sort(key=lambda r: -r[2])
print("%-40s %10s %10s" % ("param", "cpu32", "hip32"))
for r in rows[:12]: print("%-40s %10.3e %10.3e" % r)
print("median cpu32 %.3e hip %.3e" % (sorted(r[1] for r in rows)[len(rows)//2], sorted(r[2] for r in rows)[len(rows)//2]))
