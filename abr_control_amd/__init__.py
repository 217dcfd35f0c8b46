"""abr_control_amd - MI355X-native batched operational-space control.

Drop-in for the hot path of abr/abr_control: `arms.<arm>.Config` offers the reference's
`robot_config` API and `controllers.{OSC,Sliding,Joint,Damping,RestingConfig}` its
controllers; both evaluate one state or a batch of states with hand-written HIP kernels
(abr_control_amd/csrc) through the C ABI of include/abrk.h.  No CPU fallback.
"""
from . import arms, controllers  # noqa: F401
from ._lib import AbrkError, DeviceArray, Event, Stream, device_count, device_name, scratch_stats  # noqa: F401

__version__ = "0.1.0"
