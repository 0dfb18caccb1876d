#!/usr/bin/env python
"""One rank of tests/test_host_logic.py::test_bench_line_is_the_last_line_on_stdout: a gloo group, a "banner" written through C stdio
(it stays in libc's buffer on a pipe, as RCCL's version banner does), some Python output, then bench.emit_line."""
import ctypes
import os
import sys

import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
libc = ctypes.CDLL(None)
libc.printf(b"BANNER of rank %d : buffered by libc\n", rank)       # no fflush: on a pipe this would surface at exit()
print("python text of rank %d" % rank)
bench.emit_line({"metric": "m", "value": 1.0, "rank_count": world}, rank, True)
if rank != 0:
    libc.printf(b"LATE text of rank %d\n", rank)                   # a non-zero rank's stdout goes nowhere after the line
