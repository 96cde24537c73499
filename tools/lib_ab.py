#!/usr/bin/env python3
"""Run a tool against an alternative build of the library: python tools/lib_ab.py <lib.so> <tool.py> [args...]
(box-to-box variance is ~10 %, so old/new builds have to be compared on the same box, one process after the other)."""
import os, sys, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pero_pretraining_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
