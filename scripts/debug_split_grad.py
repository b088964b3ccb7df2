"""Why is the split path's gradient p90 2x torch-CPU's?  Same weights (seed 77), same 1x5x512x512
batch, fp64 oracle as ground truth: per-tensor relative max-norm errors of (A) the whole-model HIP path, (B) the HIP split path
(encoder / decoder / head called separately, nothing added in between), (C) torch-CPU fp32, and the same three on a
2x5x160x128 batch."""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "flair-1_amd")]
import torch, torch.nn as nn
import flair_amd
from oracle import unet_resnet34 as om

dev = torch.device("cuda:0")
q = lambda d, f: sorted(d.values())[min(len(d) - 1, int(f * len(d)))]


def errs(params, truth):
    return {k: float((p.grad.detach().cpu().double() - truth[k].grad).abs().max() / (truth[k].grad.abs().max() + 1e-12)) for k, p in params}


for shape in ((1, 512, 512), (2, 160, 128)):
    ref = om.seeded_model(5, 13, 77)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(shape[0], 5, shape[1], shape[2], generator=g)
    lab = torch.randint(0, 13, shape, generator=g)
    ref64 = copy.deepcopy(ref).double().train()
    nn.functional.cross_entropy(ref64(x.double()), lab).backward()
    p64 = dict(ref64.named_parameters())
    ref.train()
    nn.functional.cross_entropy(ref(x), lab).backward()
    e_cpu = errs(ref.named_parameters(), p64)
    res = {}
    for mode in ("whole", "split"):
        hip = flair_amd.create_model("unet", "resnet34", encoder_weights=None, in_channels=5, classes=13, compute_dtype="f32")
        hip.load_state_dict(om.seeded_model(5, 13, 77).state_dict())
        hip = hip.to(dev).train()
        if mode == "whole":
            out = hip(x.to(dev))
        else:
            feats = hip.encoder(x.to(dev))
            out = hip.segmentation_head(hip.decoder(*feats))
        nn.functional.cross_entropy(out, lab.to(dev)).backward()
        res[mode] = errs(hip.named_parameters(), p64)
    print(shape, "cpu   median %.3e p90 %.3e max %.3e" % (q(e_cpu, .5), q(e_cpu, .9), max(e_cpu.values())))
    for mode, e in res.items():
        print(shape, "%-5s median %.3e (%.2fx) p90 %.3e (%.2fx) max %.3e (%.2fx)" % (mode, q(e, .5), q(e, .5) / q(e_cpu, .5), q(e, .9), q(e, .9) / q(e_cpu, .9),
                                                                                 max(e.values()), max(e.values()) / max(e_cpu.values())))
    # where do whole and split differ from each other?
    worst = sorted(((res["split"][k] / (res["whole"][k] + 1e-12), k) for k in res["split"]), reverse=True)[:6]
    print("  largest split/whole ratios:", [(round(r, 2), k) for r, k in worst])
    for L_ in ("encoder.conv1", "encoder.layer1", "encoder.layer2", "encoder.layer3", "encoder.layer4", "decoder", "segmentation_head"):
        ks = [k for k in e_cpu if k.startswith(L_)]
        md = lambda e: sorted(e[k] for k in ks)[len(ks) // 2]
        print("  %-18s median cpu %.2e whole %.2e split %.2e" % (L_, md(e_cpu), md(res["whole"]), md(res["split"])))
