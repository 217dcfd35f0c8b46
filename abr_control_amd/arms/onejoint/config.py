"""One-link arm (abr_control/arms/onejoint/config.py:30-42).  As in the reference only
link0 (which does not move) is summed into M and g (N_LINKS = 1), so M is identically zero."""
import numpy as np

from ... import _abi
from ..base_config import BatchedConfig


class Config(BatchedConfig):
    def __init__(self, **kwargs):
        super().__init__(_abi.load_table("onejoint"), builtin="onejoint", **kwargs)
        self.JOINT_NAMES = ["joint0"]
        self.START_ANGLES = np.array([np.pi / 2.0])
