#!/usr/bin/env python3
"""`python train.py …`: the training harness of the HIP path (same command line as the reference's train.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import dpd_amd  # noqa: E402,F401
from dpd_amd.train import main  # noqa: E402

if __name__ == "__main__":
    main()
