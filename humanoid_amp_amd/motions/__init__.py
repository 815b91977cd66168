"""AMP motion loader and motion clips (npz; schema as the reference's motions/README.md:11-21)."""

import os

from .motion_loader import MotionLoader

MOTIONS_DIR = os.path.dirname(os.path.abspath(__file__))

__all__ = ["MotionLoader", "MOTIONS_DIR"]
