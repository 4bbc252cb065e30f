#!/usr/bin/env python3
"""Registers / LDS / scratch of every kernel instantiation of a built library, from its code object's metadata note:
    python scripts/kernel_meta.py [lib.so] [substring]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_abi_and_host import kernel_metadata  # noqa: E402

lib = sys.argv[1] if len(sys.argv) > 1 else None
pat = sys.argv[2] if len(sys.argv) > 2 else ""
md = kernel_metadata(lib)
names = sorted(k for k in md if pat in k)
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
for k, d in zip(names, dem):
    v = md[k]
    d = d.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    print(f"{d:60s} vgpr {v['vgpr_count']:>4s} sgpr {v['sgpr_count']:>4s} lds {v['group_segment_fixed_size']:>7s} "
          f"scratch {v['private_segment_fixed_size']:>4s} spill {v['vgpr_spill_count']}")
