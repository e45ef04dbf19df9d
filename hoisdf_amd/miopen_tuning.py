"""Encoder-side tuning (the CNN encoder stays PyTorch/MIOpen, BASELINE.json north_star).

MIOpen's default (immediate-mode heuristics) picks fp32 NHWC convolution solvers that leave ~6 ms/step on
the table for the ResNet-50 encoder at batch 32 (140.5 -> 134.1 ms/step on MI355X); its search
("find", ``torch.backends.cudnn.benchmark``) costs ~160 s on a fresh machine.  ``hoisdf_amd/miopen_db``
holds the user find-db / perf-db text files that search produced for the bench shapes on gfx950; this module
points MIOpen at a private writable copy so the search result is reused (MIOpen re-searches only
shapes that are missing).  fp32 everywhere; no precision change."""
from __future__ import annotations

import glob
import os
import shutil
import tempfile

_DB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_db")


def enable(find_mode: str = "3") -> str | None:
    """Call before the first convolution runs.  Returns the db directory in use (None if left alone)."""
    import torch
    if os.environ.get("MIOPEN_USER_DB_PATH"):          # the user manages MIOpen's db themselves
        torch.backends.cudnn.benchmark = True
        return os.environ["MIOPEN_USER_DB_PATH"]
    files = glob.glob(os.path.join(_DB_DIR, "*.txt"))
    if not files:
        return None
    # one writable copy per (user, local rank): ranks never share a db file, and a second process on the same
    # machine reuses what the first one had to search for (shapes missing from the shipped files)
    dst = os.path.join(tempfile.gettempdir(), f"hoisdf_miopen_db_{os.getuid()}_r{os.environ.get('LOCAL_RANK', '0')}")
    os.makedirs(dst, exist_ok=True)
    for f in files:
        if not os.path.exists(os.path.join(dst, os.path.basename(f))):
            shutil.copy(f, dst)
    os.environ["MIOPEN_USER_DB_PATH"] = dst
    os.environ.setdefault("MIOPEN_FIND_MODE", find_mode)  # 3 = hybrid: db hit -> no search
    torch.backends.cudnn.benchmark = True
    return dst
