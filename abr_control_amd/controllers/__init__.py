"""Controllers with the reference's API (abr_control/controllers/__init__.py) whose
generate() runs on the GPU for one state or a batch.

Every class keeps the constructor arguments, attribute names, return types and exceptions of its namesake; (n,) inputs
give the reference's shapes, (B, n) inputs (NumPy or DeviceArray) give [B, ...].  OSC fuses Damping / RestingConfig into
its kernel and sums the other secondary controllers on the device; there is no host implementation of any law."""
from .avoid_joint_limits import AvoidJointLimits
from .avoid_obstacles import AvoidObstacles
from .controller import Controller
from .damping import Damping
from .floating import Floating
from .joint import Joint
from .osc import OSC
from .resting_config import RestingConfig
from .sliding import Sliding

__all__ = ["AvoidJointLimits", "AvoidObstacles", "Controller", "Damping", "Floating", "Joint", "OSC", "RestingConfig",
           "Sliding"]
