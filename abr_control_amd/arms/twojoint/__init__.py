"""Planar two-link arm: constant table of abr_control/arms/twojoint/config.py; ArmSim = the batched plant.

Nothing is generated or compiled per arm at run time: `Config()` only registers the table with libabrk.so."""
from .arm_sim import ArmSim
from .config import Config

__all__ = ['ArmSim', 'Config']
