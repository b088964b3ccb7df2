import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flair-1_amd")]
import torch
import flair_amd
from oracle import segformer as osf
dev = torch.device("cuda:0")
for dt in ("f32", "bf16"):
    for shape in ((1, 128, 128), (2, 256, 256), (1, 512, 512)):
        ref = osf.seeded_model()
        hip = flair_amd.SegformerForSemanticSegmentation(num_channels=5, num_labels=19, compute_dtype=dt)
        hip.load_state_dict(ref.state_dict())
        hip = hip.to(dev)
        x = torch.randn(shape[0], 5, shape[1], shape[2], generator=torch.Generator().manual_seed(3))
        lq, lf = osf.logits(ref, x)
        gq = hip(x.to(dev)).logits.cpu()
        gf = hip.forward_full(x.to(dev)).cpu()
        sc = float(lq.abs().max())
        print(dt, shape, "scale %.3f quarter max|d| %.3e full max|d| %.3e rms %.3e flips %d / %d" % (
            sc, float((gq - lq).abs().max()), float((gf - lf).abs().max()), float((gq - lq).pow(2).mean().sqrt()),
            int((gf.argmax(1) != lf.argmax(1)).sum()), lf.argmax(1).numel()), flush=True)
