"""CPU oracle for the FLAIR-1 segmentation hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it; the product path (``flair-1_amd/``) never does and
fails loudly when its HIP library is missing.

Contents
--------
* ``unet_resnet34.py``  torch-CPU fp32 restatement of the third-party model the
  reference instantiates at ``src/flair/model.py:37-41``
  (``segmentation-models-pytorch==0.3.3`` ``Unet('resnet34')``, pinned at
  ``setup.py:36``; it is NOT under /root/reference and not installed here).
* ``seg_step.py``       restatement of ``segmentation_task_training.step``
  (``src/flair/task_module.py:65-80``), the criterion
  (``src/flair/tasks_utils.py:88-93``), the torchmetrics-1.2.0 confusion
  matrix / Jaccard semantics (``task_module.py:36-51``) and the offline IoU
  family (``src/flair/metrics.py:10-40``), in numpy / plain torch.

Pinning status
--------------
* IoU / OA / precision / recall / F-score: PINNED.  ``tests/golden/make_golden.py``
  imports the reference's own ``src/flair/metrics.py`` (with a one-line stub for
  ``rank_zero_only``) and stores its outputs in ``tests/golden/metrics_*.json``;
  SURVEY.md §4 known answer (mIoU 56.541990939912836) is reproduced.
* step()/predict_step() control flow: PINNED against the reference's own
  ``model.py`` / ``task_module.py`` run in-process on this oracle's model
  (``tests/golden/make_golden.py``).
* The smp U-Net arithmetic itself (third-party, absent, and the reference has
  no tests or golden outputs for it): PARITY UNPINNED except for the
  known-answer parameter counts 24,436,369 / 24,444,381 / 24,445,251 and the
  state-dict key list.
"""
