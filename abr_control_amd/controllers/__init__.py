"""Controllers with the reference's API (abr_control/controllers/__init__.py) whose
generate() runs on the GPU for one state or a batch."""
from .avoid_joint_limits import AvoidJointLimits
from .avoid_obstacles import AvoidObstacles
from .controller import Controller
from .damping import Damping
from .floating import Floating
from .joint import Joint
from .osc import OSC
from .resting_config import RestingConfig
from .sliding import Sliding
