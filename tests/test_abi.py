"""CPU-side checks of the C-ABI library: it loads and exports every symbol include/msseg.h declares."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "msseg.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(msseg_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from medicalsemseg_amd import hip
    if not os.path.exists(hip.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = hip.load_library()
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/msseg.h but not exported"
        assert n in hip.SIGNATURES, f"{n} has no ctypes signature in hip.py"
    assert lib.msseg_abi_version() == 1
    assert lib.msseg_cout_block(32) == 32 and lib.msseg_cout_block(48) == 48 and lib.msseg_cout_block(3) == 16


def test_product_refuses_cpu():
    import torch
    from medicalsemseg_amd.losses import DiceCELoss
    from medicalsemseg_amd.models.unet import UNet
    net = UNet(1, 2, (16, 16, 32, 64, 128, 16))
    with pytest.raises(RuntimeError, match="GPU only"):
        net((torch.zeros(1, 1, 16, 16, 16), None, None))
    with pytest.raises(RuntimeError, match="GPU only"):
        DiceCELoss()(torch.zeros(1, 2, 4, 4, 4), torch.zeros(1, 1, 4, 4, 4))


def test_state_dict_layout_matches_monai_names():
    from medicalsemseg_amd.models.unet import UNet
    from oracle.blocks import BasicUNet
    a = UNet(1, 3).state_dict()
    b = BasicUNet(1, 3).state_dict()
    assert list(a.keys()) == list(b.keys())
    assert all(a[k].shape == b[k].shape for k in a)
    assert sum(v.numel() for v in a.values()) == 5749443  # BasicUNet 1->3
