#!/usr/bin/env python3
"""Run pytest with the device allocator's free memory filled with NaN first (test tool): kernels that read a buffer they
were supposed to write first -- torch.empty workspaces, padded tile rows -- see NaN instead of whatever finite bytes the last
process left, so `0 * garbage` patterns show up as failures.  python tools/poison_run.py <pytest args>"""
import sys
import torch

big = torch.full((1 << 29,), float("nan"), device="cuda")           # 2 GiB of the large pool
small = [torch.full((1 << 16,), float("nan"), device="cuda") for _ in range(2048)]   # 512 MiB of 256 KiB blocks
tiny = [torch.full((1 << 10,), float("nan"), device="cuda") for _ in range(4096)]    # small pool
torch.cuda.synchronize()
del big, small, tiny
import pytest
sys.exit(pytest.main(sys.argv[1:]))
