#!/usr/bin/env python3
"""Drop-in for the reference's ./dmp.py as monitorer-sim.py spawns it with model='dmp'
(monitorer-sim.py:30-31, 229-236): same argv, same label / initial-* files, exit code 0."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gnode.trainer import main_dmp  # noqa: E402

if __name__ == "__main__":
    sys.exit(main_dmp())
