"""Planar three-link arm: constant table of abr_control/arms/threejoint/config.py.

Nothing is generated or compiled per arm at run time: `Config()` only registers the table with libabrk.so."""
from .config import Config

__all__ = ['Config']
