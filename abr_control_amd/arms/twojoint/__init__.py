from .arm_sim import ArmSim
from .config import Config
