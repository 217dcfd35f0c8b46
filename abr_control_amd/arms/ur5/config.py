"""UR5 (6 joints).  Mirrors abr_control/arms/ur5/config.py:10-47 (attributes) - the frame
chain itself lives in arms/tables/ur5.json / csrc/abrk_arms_builtin.h."""
import numpy as np

from ... import _abi
from ..base_config import BatchedConfig


class Config(BatchedConfig):
    def __init__(self, **kwargs):
        super().__init__(_abi.load_table("ur5"), builtin="ur5", **kwargs)
        self.JOINT_NAMES = [f"UR5_joint{ii}" for ii in range(self.N_JOINTS)]
        self.START_ANGLES = np.array(
            [0, np.pi / 4.0, -np.pi / 2.0, np.pi / 4.0, np.pi / 2.0, np.pi / 2.0], dtype="float32")
