#!/usr/bin/env python
"""Tile configuration of every igemm op of the segment plan at the given batch sizes (host only: the plan builder
of csrc/plan.cpp through the CPU interpreter library of tests/). Usage: plan_dump.py [batch ...] > file"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demucs_cpp_amd.weights import write_synthetic_model  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, "tests", "_build", "libcpu_interp.so"))
lib.interp_create_plan.restype = ctypes.c_void_p
lib.interp_create_plan.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_int]
lib.interp_plan_dump.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
path = "/tmp/plan_dump_4s.bin"
if not os.path.exists(path):
    write_synthetic_model(path, 4, 0)
for b in [int(x) for x in sys.argv[1:]] or [1, 2, 4, 12, 24]:
    h = lib.interp_create_plan(path.encode(), 343980, b)
    buf = ctypes.create_string_buffer(1 << 18)
    assert lib.interp_plan_dump(h, buf, 1 << 18) > 0
    for ln in buf.value.decode().splitlines():
        print(b, ln)
