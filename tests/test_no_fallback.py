"""The product path has no eager-PyTorch / CPU form of a hot-path reduction (a CPU tensor raises in
hoisdf_amd.ops, a missing library raises HoisdfLibraryError).  Round 4's verdict found four `else` branches that computed a loss with
torch.nn.functional when the tensors were not on the device; they are gone, and this keeps them gone."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _src(rel):
    return open(os.path.join(ROOT, rel)).read()


def test_no_functional_loss_fallback_in_the_hot_path_modules():
    for rel in ("hoisdf_amd/model.py", "hoisdf_amd/nets/heads.py", "hoisdf_amd/nets/blocks.py"):
        s = _src(rel)
        for name in ("F.smooth_l1_loss", "F.l1_loss", "F.binary_cross_entropy", "F.grid_sample", "F.layer_norm", "F.softmax",
                     "F.multi_head_attention_forward", "F.linear"):
            assert name not in s, (rel, name)
        # no "if x.is_cuda: HIP else: torch" dispatch
        assert not re.search(r"\.is_cuda\s*:", s), rel


def test_ops_refuse_cpu_tensors():
    from hoisdf_amd import ops
    with pytest.raises(RuntimeError):
        ops._chk(torch.zeros(4, 4))


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "hoisdf_amd")):
        for f in files:
            if f.endswith(".py"):
                s = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", s, re.M), os.path.join(dirpath, f)
