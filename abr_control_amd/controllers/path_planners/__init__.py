"""Path planners on the batched engine.  Only the planner that sits on the hot path's callers is here:
InverseKinematics (abr_control/controllers/path_planners/inverse_kinematics.py:28-135), all iterations of a path in one
kernel.  The reference's profile-based planners are host-side, run once per movement, and work unchanged on top of the
batched robot_config."""
from .inverse_kinematics import InverseKinematics

__all__ = ["InverseKinematics"]
