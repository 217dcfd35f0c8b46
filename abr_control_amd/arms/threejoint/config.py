"""Planar three-link arm (abr_control/arms/threejoint/config.py:32-68)."""
import numpy as np

from ... import _abi
from ..base_config import BatchedConfig


class Config(BatchedConfig):
    def __init__(self, **kwargs):
        super().__init__(_abi.load_table("threejoint"), builtin="threejoint", **kwargs)
        self.START_ANGLES = np.array([np.pi / 4.0, np.pi / 4.0, np.pi / 4.0], dtype="float32")
