"""One-joint arm: constant table of abr_control/arms/onejoint/config.py.

Nothing is generated or compiled per arm at run time: `Config()` only registers the table with libabrk.so."""
from .config import Config

__all__ = ['Config']
