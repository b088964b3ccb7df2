#!/usr/bin/env python3
"""Fixture of the SegFormer row (BASELINE config 5) from the ORACLE (oracle/segformer.py = transformers'
SegformerForSemanticSegmentation on the CPU, fp32, seeded weights; parity unpinned against the reference, see there):
one 5x256x256 tile through MiT-B2 with 19 labels -> a crop of the quarter-resolution logits, their per-class mean, the
upsampled argmax mask.  Data only.

    python tests/golden/make_golden_segformer.py
"""
import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import segformer as osf  # noqa: E402

torch.set_num_threads(8)
m = osf.seeded_model(5, 19, seed=2022)
x = torch.randn(1, 5, 256, 256, generator=torch.Generator().manual_seed(17))
lq, lf = osf.logits(m, x)
mask = lf.argmax(1).numpy().astype(np.uint8)
np.savez_compressed(os.path.join(HERE, "segformer_b2_c19.npz"), model_seed=np.int64(2022), tile_seed=np.int64(17),
                    logits_quarter_crop=lq[0, :, 24:40, 24:40].numpy(), logits_quarter_mean=lq.double().mean(dim=(0, 2, 3)).numpy(),
                    logits_full_crop=lf[0, :, 100:132, 100:132].numpy(), mask=mask,
                    mask_sha256=np.array(hashlib.sha256(mask.tobytes()).hexdigest()))
print("segformer golden:", lq.shape, float(lq.abs().max()))
