"""Kinova Jaco2 (6 joints as the reference defines it, abr_control/arms/jaco2/config.py:37-38,
hand COM as link6, EE at the finger tips)."""
import numpy as np

from ... import _abi
from ..base_config import BatchedConfig


class Config(BatchedConfig):
    def __init__(self, **kwargs):
        super().__init__(_abi.load_table("jaco2"), builtin="jaco2", **kwargs)
        self.JOINT_NAMES = [f"joint{ii}" for ii in range(self.N_JOINTS)]
        self.START_ANGLES = np.array([2.0, 3.14, 1.57, 4.71, 0.0, 3.04])
