"""Per-tensor cosine similarity between the bf16 and fp32 HIP gradients of one step (debug aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "flair-1_amd"))
import torch
import flair_amd
from oracle import unet_resnet34 as om

dev = torch.device("cuda:0")
B, H, W = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (3, 96, 160))]
ref = om.seeded_model(5, 13, 77)
g = torch.Generator().manual_seed(H * W)
x = torch.randn(B, 5, H, W, generator=g).to(dev)
lab = torch.randint(0, 13, (B, H, W), generator=g).to(torch.uint8).to(dev)
grads = {}
for name in ("f32", "bf16"):
    m = flair_amd.create_model("unet", "resnet34", encoder_weights=None, in_channels=5, classes=13, compute_dtype=name)
    m.load_state_dict(ref.state_dict(), strict=True)
    m = m.to(dev).train()
    tr = flair_amd.SegTrainer(m, lr=0.0)
    print(name, "loss", tr.train_step(x, lab).item())
    grads[name] = {k: v.clone() for k, v in zip([n for n, _ in m.named_parameters()], m._grad_views(tr.grads))}
worst = []
for k in grads["f32"]:
    a, b = grads["f32"][k].double().flatten(), grads["bf16"][k].double().flatten()
    cos = float(torch.nn.functional.cosine_similarity(a, b, dim=0))
    worst.append((cos, k, float(a.norm()), float(b.norm())))
for cos, k, na, nb in sorted(worst)[:25]:
    print(f"{cos:7.3f}  |g32|={na:9.3e} |g16|={nb:9.3e}  {k}")
print("median cos", sorted(w[0] for w in worst)[len(worst) // 2])
