#!/usr/bin/env python3
"""Entry point mirroring the reference's gen_diverse_grasp_obman.py (generation plumbing only; metrics are out of scope).
Run from the repository root:  python d-vqvae_amd/gen_diverse_grasp_obman.py [--num_grasp N] [--objects cloud.npy ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dvqvae_amd  # noqa: E402,F401
from dvqvae_amd.generate import main  # noqa: E402

if __name__ == "__main__":
    main("obman")
