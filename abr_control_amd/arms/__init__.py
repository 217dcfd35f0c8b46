"""Arm configs with the reference's `robot_config` API (abr_control/arms)."""
from . import jaco2, onejoint, threejoint, twojoint, ur5  # noqa: F401
from .base_config import BatchedConfig  # noqa: F401


def from_table(table, **kwargs):
    """robot_config for a user arm table (tools/extract_arm_table.py): the runtime-table kernels, or - `compiled=True`,
    or a plugin already in the cache - kernels specialised for the table (abr_control_amd/specialize.py)."""
    return BatchedConfig(table, builtin=None, **kwargs)
