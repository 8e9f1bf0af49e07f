#!/usr/bin/env python3
"""CPU baseline probe: `cpu_step.py [sample_div] [threads]` (see bench.cpu_baseline_step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import cpu_baseline_step
div = int(sys.argv[1]) if len(sys.argv) > 1 else 4
thr = int(sys.argv[2]) if len(sys.argv) > 2 else None
print(cpu_baseline_step(800, 1333, div, thr))
