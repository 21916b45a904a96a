#!/usr/bin/env python3
"""Drop-in for the reference's ./ode_nn_ngraphs.py as monitorer-ngraphs.py spawns it
(monitorer-ngraphs.py:25-30, 131) with model='ode_nn': same argv, same files, exit code 0."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gnode.trainer import main_multi  # noqa: E402

if __name__ == "__main__":
    sys.exit(main_multi())
