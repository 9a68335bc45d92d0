#!/usr/bin/env python3
"""python -m lightweaver_amd.spill_lint from anywhere (the check itself lives in the package: the build runs it)."""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
runpy.run_module('lightweaver_amd.spill_lint', run_name='__main__')
