"""Controllers with the reference's API (abr_control/controllers/__init__.py) whose
generate() runs on the GPU for one state or a batch."""
from .controller import Controller
from .damping import Damping
from .joint import Joint
from .osc import OSC
from .resting_config import RestingConfig
from .sliding import Sliding
