"""Planar two-link arm (abr_control/arms/twojoint/config.py:30-63)."""
import numpy as np

from ... import _abi
from ..base_config import BatchedConfig


class Config(BatchedConfig):
    def __init__(self, **kwargs):
        super().__init__(_abi.load_table("twojoint"), builtin="twojoint", **kwargs)
        self.START_ANGLES = np.array([np.pi / 4.0, np.pi / 4.0])
