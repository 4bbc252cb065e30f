#!/usr/bin/env python3
"""Does a candidate block behind a TRANSIENT gap (placement.FAR_GAPS_BYTES) land in another 96 GiB class?  The capped search is cut
short on purpose (max_hold_bytes = the arrays' own blocks), so that the far candidates are what is probed; fresh process per run."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gobblet_rl_amd as G  # noqa: E402

if os.environ.get("GOBBLET_HIP_LIB"):  # an experiment's own build of the library (scripts/build_variant.sh)
    G._native.use_library(os.environ["GOBBLET_HIP_LIB"])
from gobblet_rl_amd import placement  # noqa: E402

n, T = 1 << 20, 20
dev = torch.device("cuda:0")
cells = T * n
a, b, info = placement.spread_pair(cells * 117, cells * 54, dev, slot_boards=n, plies=T, max_hold_bytes=1,
                                   plain=lambda: (torch.empty(cells * 117, dtype=torch.uint8, device=dev),
                                                  torch.empty(cells * 54, dtype=torch.uint8, device=dev)))
print(info)
