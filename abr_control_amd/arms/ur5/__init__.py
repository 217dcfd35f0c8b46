"""UR5 (6 joints): constant table of abr_control/arms/ur5/config.py evaluated by the HIP kernels.

Nothing is generated or compiled per arm at run time: `Config()` only registers the table with libabrk.so."""
from .config import Config

__all__ = ['Config']
