"""BASELINE.json configs[4] at full size on one MI355X: `python bench.py --config c5` (ByT5-base encoder over 256 states +
masked top-100 over 1,000,000 x 1536 premises held as an e4m3 index; the bf16-index scan beside it).  This file only
forwards to it - the measurement lives in bench.py (run_c5) since round 6, so that a driver-run command reports it.
    python tools/c5_bench.py [--steps 5]
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.exit(subprocess.call([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c5", *sys.argv[1:]]))
