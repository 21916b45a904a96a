#!/usr/bin/env python3
"""Drop-in for the reference's ./ode_nn_ngraph_sim.py as monitorer-sim.py spawns it
(monitorer-sim.py:26-33, 229-236) with model='ode_nn': same argv, same files, exit code 0."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gnode.trainer import main_single  # noqa: E402

if __name__ == "__main__":
    sys.exit(main_single())
