"""CPU oracle for the 3-D segmentation hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / reported baseline.

The oracle is a plain PyTorch-CPU fp32 restatement of the reference's
arithmetic (stock ``torch`` ops composed exactly the way the reference and the
MONAI blocks it calls compose them).  Pinning status, per component:

* Swin encoder (``oracle/swin.py``): pinned against golden vectors produced by
  importing the reference's own ``models/backbones/swin_nnformer.py`` in the
  build container (``oracle/gen_golden.py`` -> ``tests/golden/*.npz``).
* UNETRC conv/deconv decoder blocks, LR schedule, affine helpers: pinned the
  same way (reference modules import with stock torch only).
* MONAI-defined pieces (UnetResBlock / UnetrUpBlock / UnetOutBlock, BasicUNet,
  DiceCELoss, DiceMetric, dense_patch_slices, compute_importance_map): MONAI is
  not vendored in the reference and not installable here -> **parity unpinned**
  against MONAI itself; restated from its published algorithm, anchored on the
  reference's call sites and on closed-form known-answer tests.
"""
