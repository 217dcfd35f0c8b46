"""Operational space controller, same constructor / generate() as
abr_control/controllers/osc.py:8-320, evaluated for one state or a batch by one fused HIP
kernel (forward kinematics, J, M, g[, C], task-space inertia, control law, null space)."""
import numpy as np

from .. import _abi, engine
from .._lib import DeviceArray
from .controller import Controller
from .damping import Damping
from .resting_config import RestingConfig


class OSC(Controller):
    def __init__(self, robot_config, kp=1, ko=None, kv=None, ki=0, vmax=None, ctrlr_dof=None,
                 null_controllers=None, use_g=True, use_C=False, orientation_algorithm=0):
        super().__init__(robot_config)
        self._require_batched_config()
        # osc.py:70-90
        self.kp = kp
        self.ko = kp if ko is None else ko
        self.kv = np.sqrt(self.kp + self.ko) if kv is None else kv
        self.ki = ki
        self.null_controllers = null_controllers
        self.use_g = use_g
        self.use_C = use_C
        self.orientation_algorithm = orientation_algorithm
        if orientation_algorithm not in (0, 1):
            raise Exception(  # raised lazily by the reference (osc.py:190-194)
                f"Invalid algorithm number {orientation_algorithm} for calculating orientation error")
        if self.ki != 0:
            self.integrated_error = np.zeros(6)
        if ctrlr_dof is None:
            ctrlr_dof = [True, True, True, False, False, False]
        self.ctrlr_dof = np.copy(ctrlr_dof)
        self.n_ctrlr_dof = np.sum(self.ctrlr_dof)
        self.task_space_gains = np.array([self.kp] * 3 + [self.ko] * 3)
        self.lamb = self.task_space_gains / self.kv
        if self.n_ctrlr_dof > robot_config.N_JOINTS:  # osc.py:93-99
            print(f"\nRobot has fewer DOF ({robot_config.N_JOINTS}) than the specified number of "
                  f"space dimensions to control ({self.n_ctrlr_dof}), Poor performance may result.\n")
        self.vmax = vmax
        if vmax is not None:  # osc.py:110-115
            self.sat_gain_xyz = vmax[0] / self.kp * self.kv
            self.sat_gain_abg = vmax[1] / self.ko * self.kv
            self.scale_xyz = vmax[0] / self.kp * self.kv
            self.scale_abg = vmax[1] / self.ko * self.kv
        self.ZEROS_SIX = np.zeros(6)
        self.IDENTITY_N_JOINTS = np.eye(self.robot_config.N_JOINTS)
        self.training_signal = None

        # secondary controllers: Damping / RestingConfig are fused into the kernel; any other
        # object with generate(q, dq) is evaluated by the caller's code and only projected
        self._fused, self._foreign = [], []
        for nc in null_controllers or []:
            if type(nc) is Damping:
                self._fused.append(_abi.make_damping(nc.kv))
            elif type(nc) is RestingConfig:
                self._fused.append(_abi.make_resting(nc.rest_angles_list, nc.kp, nc.kv))
            else:
                self._foreign.append(nc)

    def _params(self, ref_frame, xyz_offset):
        rc = self.robot_config
        return _abi.make_osc_params(
            rc.N_JOINTS, kp=self.kp, ko=self.ko, kv=self.kv, ki=self.ki, vmax=self.vmax,
            ctrlr_dof=self.ctrlr_dof, null_controllers=self._fused, use_g=self.use_g, use_C=self.use_C,
            orientation_algorithm=self.orientation_algorithm, ref_frame=ref_frame, xyz_offset=xyz_offset)

    def generate(self, q, dq, target, target_velocity=None, ref_frame="EE", xyz_offset=None):
        """Control signal(s) moving `ref_frame` to `target` (osc.py:217-320).

        q, dq: (n,) or (B, n); target: (6,) or (B, 6) [x,y,z,alpha,beta,gamma];
        target_velocity: None, (6,) or (B, 6).  Returns float64 (n,) or (B, n)
        (kernel dtype for a float32 config / DeviceArrays)."""
        rc = self.robot_config
        params = self._params(ref_frame, xyz_offset)
        (q2, dq2, t2, tv2), single = self._rows(q, dq, target, target_velocity)
        on_device = isinstance(q2, DeviceArray)
        B = q2.shape[0]
        ie = None
        if self.ki != 0:  # per-row state (osc.py:81-82, 262-264)
            ie = self.integrated_error
            if on_device:
                if not isinstance(ie, DeviceArray) or ie.shape != (B, 6):
                    ie = self.integrated_error = DeviceArray((B, 6), rc.dtype, rc.device).zero_()
            else:
                ie = np.asarray(ie, dtype=rc.dtype)
                if ie.shape == (6,) and B == 1:
                    ie = ie.reshape(1, 6)
                if ie.shape != (B, 6):
                    ie = np.zeros((B, 6), rc.dtype)
                ie = np.ascontiguousarray(ie)
        une = None
        if self._foreign:
            if on_device:
                raise TypeError("Python null controllers need NumPy states (they are evaluated on the host)")
            une = np.zeros((B, rc.N_JOINTS), rc.dtype)
            for nc in self._foreign:
                for b in range(B):
                    une[b] += nc.generate(q2[b], dq2[b])
        u, ts = engine.osc_generate(rc.arm_id, rc.N_JOINTS, params, q2, dq2, t2, tv2, ie, une,
                                    training_signal=True, dtype=rc.dtype, device=rc.device)
        if on_device:
            self.training_signal = ts
            return u
        if self.ki != 0:
            self.integrated_error = ie[0] if single else ie
        if rc.reference_dtypes:
            u, ts = u.astype(np.float64), ts.astype(np.float64)
        self.training_signal = ts[0] if single else ts
        return u[0] if single else u
