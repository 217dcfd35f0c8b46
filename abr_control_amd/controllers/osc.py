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
        from ..arms.base_config import BatchedConfig

        # A foreign (duck-typed) robot_config - e.g. the reference's MujocoConfig, whose J/M/g/Tx/R are
        # read from the simulator - keeps working: its methods are called per state on the host exactly as
        # osc.py:242-301 does, and the control law itself runs on the GPU (abrk_osc_law_batch).
        self._fused_config = isinstance(robot_config, BatchedConfig)
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
        self._params_cache = {}

        # secondary controllers: Damping / RestingConfig are fused into the kernel; the package's other
        # controllers (AvoidJointLimits, AvoidObstacles, Floating) are summed on the device by their own
        # kernels and enter the null-space filter as u_null_ext; any other object with generate(q, dq)
        # is evaluated by the caller's code per state and only projected
        # (_fused_src holds the live controller objects: their gains are read at every generate(), as the
        #  reference calls null_controller.generate() every tick, osc.py:311-313 - gain scheduling keeps working)
        self._fused_src, self._device, self._foreign = [], [], []
        for nc in null_controllers or []:
            if not self._fused_config:
                self._foreign.append(nc)  # evaluated with the foreign config's own M(q)
            elif type(nc) in (Damping, RestingConfig) and len(self._fused_src) < _abi.MAX_NULL:
                self._fused_src.append(nc)
            elif hasattr(nc, "_accumulate") and getattr(nc, "robot_config", None) is robot_config:
                self._device.append(nc)  # also Damping / RestingConfig beyond the MAX_NULL fused slots
            else:
                self._foreign.append(nc)

    def _fused_key(self):
        """current gains of the fused secondary controllers (part of the parameter-cache key)"""
        key = []
        for nc in self._fused_src:
            if type(nc) is Damping:
                key.append(("d", float(nc.kv)))
            else:
                key.append(("r", float(nc.kp), float(nc.kv),
                            tuple(None if a is None else float(a) for a in nc.rest_angles_list)))
        return tuple(key)

    @property
    def _fused(self):
        return [_abi.make_damping(nc.kv) if type(nc) is Damping else _abi.make_resting(nc.rest_angles_list, nc.kp, nc.kv)
                for nc in self._fused_src]

    def _params(self, ref_frame, xyz_offset):
        """abrk_osc_params of the current attribute values (cached: a control loop calls generate() per tick)"""
        rc = self.robot_config
        if not self._fused_config:
            ref_frame, xyz_offset = "EE", None  # applied by the foreign config when it produced J/Tx/R
        key = (ref_frame, None if xyz_offset is None else tuple(float(v) for v in xyz_offset), self.kp, self.ko,
               self.kv, self.ki, None if self.vmax is None else tuple(self.vmax), tuple(self.ctrlr_dof), self.use_g,
               self.use_C, self.orientation_algorithm, self._fused_key())
        hit = self._params_cache.get(key)
        if hit is None:
            if len(self._params_cache) > 64:
                self._params_cache.clear()
            hit = self._params_cache[key] = self._make_params(ref_frame, xyz_offset)
        return hit

    def _make_params(self, ref_frame, xyz_offset):
        rc = self.robot_config
        return _abi.make_osc_params(
            rc.N_JOINTS, kp=self.kp, ko=self.ko, kv=self.kv, ki=self.ki, vmax=self.vmax,
            ctrlr_dof=self.ctrlr_dof, null_controllers=self._fused, use_g=self.use_g, use_C=self.use_C,
            orientation_algorithm=self.orientation_algorithm, ref_frame=ref_frame, xyz_offset=xyz_offset)

    # ---- the helper methods the reference's own tests call (controllers/tests/test_osc.py); each is one
    # kernel launch - nothing here is computed on the host
    def _device_dtype(self):
        rc = self.robot_config
        return getattr(rc, "dtype", np.float64), getattr(rc, "device", 0)

    def _Mx(self, M, J, threshold=1e-3):
        """Task-space inertia (osc.py:120-147): M [n,n], J [k,n] -> (Mx [k,k], M_inv [n,n]); stacks
        [B,n,n] / [B,k,n] give [B,k,k] / [B,n,n]."""
        dtype, device = self._device_dtype()
        M, J = np.asarray(M, dtype=dtype), np.asarray(J, dtype=dtype)
        single = M.ndim == 2
        M3, J3 = (M[None], J[None]) if single else (M, J)
        Mx, Minv = engine.osc_mx(M3.shape[-1], M3, J3, threshold, dtype=dtype, device=device)
        return (Mx[0], Minv[0]) if single else (Mx, Minv)

    def _calc_orientation_forces(self, target_abg, q, ref_frame):
        """Task-space orientation error (osc.py:149-196); (3,), (n,) -> (3,) or [B,3], [B,n] -> [B,3]"""
        rc = self.robot_config
        dtype, device = self._device_dtype()
        single = np.ndim(q) == 1
        # full-precision rotation (the reference goes through the float32 cast of robot_config.R, base_config.py:301)
        R = rc._eval("R", q, None, ref_frame) if self._fused_config else rc.R(ref_frame, q)
        R = np.asarray(R, dtype=dtype).reshape(-1, 3, 3)
        abg = np.ascontiguousarray(np.broadcast_to(np.atleast_2d(np.asarray(target_abg, dtype=dtype)), (R.shape[0], 3)))
        u = engine.osc_orientation_forces(self.orientation_algorithm, R, abg, dtype=dtype, device=device)
        u = u.astype(np.float64)
        return u[0] if single else u

    def _velocity_limiting(self, u_task):
        """Scale the task-space signal so that the velocity limits hold (osc.py:198-215); (6,) or [B,6]"""
        if self.vmax is None:
            raise AttributeError("'OSC' object has no attribute 'sat_gain_xyz'")  # what the reference raises
        dtype, device = self._device_dtype()
        ut = np.asarray(u_task, dtype=dtype)
        single = ut.ndim == 1
        out = engine.osc_velocity_limiting(self._params("EE", None), np.atleast_2d(ut), dtype=dtype, device=device)
        out = out.astype(np.float64)
        return out[0] if single else out

    def generate(self, q, dq, target, target_velocity=None, ref_frame="EE", xyz_offset=None, return_dynamics=None):
        """Control signal(s) moving `ref_frame` to `target` (osc.py:217-320).

        return_dynamics (extension): a subset of ("Tx", "J", "M", "g", "C", "dJ") -> returns (u, {name: array}) with the
        robot_config outputs of `ref_frame` / `xyz_offset` that the law consumed, from the same kernel launch
        (one forward kinematics instead of two; full precision, kernel dtype).

        q, dq: (n,) or (B, n); target: (6,) or (B, 6) [x,y,z,alpha,beta,gamma];
        target_velocity: None, (6,) or (B, 6).  Returns float64 (n,) or (B, n)
        (kernel dtype for a float32 config / DeviceArrays)."""
        rc = self.robot_config
        if not self._fused_config:
            if return_dynamics:
                raise TypeError("return_dynamics needs an abr_control_amd arm config (a foreign config's J/M/g are its own)")
            return self._generate_foreign(q, dq, target, target_velocity, ref_frame, xyz_offset)
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
                ie = self._host_integrated_error(B, rc.dtype)
        une = None
        if self._foreign:
            if on_device:
                raise TypeError("Python null controllers need NumPy states (they are evaluated on the host)")
            une = np.zeros((B, rc.N_JOINTS), rc.dtype)
            for nc in self._foreign:
                for b in range(B):
                    une[b] += nc.generate(q2[b], dq2[b])
        if self._device:
            if une is None:
                une = (DeviceArray((B, rc.N_JOINTS), rc.dtype, rc.device).zero_() if on_device
                       else np.zeros((B, rc.N_JOINTS), rc.dtype))
            for nc in self._device:
                nc._accumulate(q2, dq2, une)
        dyn = None
        if return_dynamics:
            u, ts, dyn = engine.osc_generate(rc.arm_id, rc.N_JOINTS, params, q2, dq2, t2, tv2, ie, une,
                                             training_signal=True, dtype=rc.dtype, device=rc.device,
                                             want=tuple(return_dynamics))
        else:
            u, ts = engine.osc_generate(rc.arm_id, rc.N_JOINTS, params, q2, dq2, t2, tv2, ie, une,
                                        training_signal=True, dtype=rc.dtype, device=rc.device)
        if on_device:
            self.training_signal = ts
            return (u, dyn) if dyn is not None else u
        if self.ki != 0:
            self.integrated_error = ie[0] if single else ie
        if rc.reference_dtypes:
            u, ts = u.astype(np.float64), ts.astype(np.float64)
        self.training_signal = ts[0] if single else ts
        if dyn is not None:
            dyn = {k: (v[0] if single else v) for k, v in dyn.items()}
            return (u[0] if single else u), dyn
        return u[0] if single else u

    def _host_integrated_error(self, B, dtype):
        """the per-row integral state (osc.py:81-82, 262-264) as the [B, 6] array a call updates in place: a single
        state keeps it as (6,) between calls (what the reference stores); a batch of another size starts from zero"""
        ie = np.asarray(self.integrated_error, dtype=dtype)
        if ie.shape == (6,) and B == 1:
            ie = ie.reshape(1, 6)
        if ie.shape != (B, 6):
            ie = np.zeros((B, 6), dtype)
        return np.ascontiguousarray(ie)

    def _generate_foreign(self, q, dq, target, target_velocity, ref_frame, xyz_offset):
        """robot_config is not an abr_control_amd config: gather its J/M/Tx/g/C/R per state (its own code,
        as osc.py:242-301 calls them) and run the law on the GPU."""
        rc = self.robot_config
        n = rc.N_JOINTS
        single = np.ndim(q) == 1
        q2, dq2 = np.atleast_2d(np.asarray(q, float)), np.atleast_2d(np.asarray(dq, float))
        B = q2.shape[0]
        t2 = np.ascontiguousarray(np.broadcast_to(np.atleast_2d(np.asarray(target, float)), (B, 6)))
        tv2 = None
        if target_velocity is not None:
            tv2 = np.ascontiguousarray(np.broadcast_to(np.atleast_2d(np.asarray(target_velocity, float)), (B, 6)))
        pos_on, ori_on = bool(np.sum(self.ctrlr_dof[:3])), bool(np.sum(self.ctrlr_dof[3:]))
        J = np.array([rc.J(ref_frame, q2[b], x=xyz_offset) for b in range(B)], dtype=float)
        M = np.array([rc.M(q2[b]) for b in range(B)], dtype=float)
        xyz = np.array([rc.Tx(ref_frame, q2[b], x=xyz_offset) for b in range(B)], dtype=float) if pos_on else None
        R = np.array([rc.R(ref_frame, q2[b]) for b in range(B)], dtype=float) if ori_on else None
        g = np.array([rc.g(q=q2[b]) for b in range(B)], dtype=float) if self.use_g else None
        Cdq = None
        if self.use_C:
            Cdq = np.array([np.dot(rc.C(q=q2[b], dq=dq2[b]), dq2[b]) for b in range(B)], dtype=float)
        une = None
        if self._foreign:
            une = np.zeros((B, n))
            for nc in self._foreign:
                for b in range(B):
                    une[b] += nc.generate(q2[b], dq2[b])
        ie = None
        if self.ki != 0:
            ie = self._host_integrated_error(B, float)
        device = getattr(rc, "device", 0)
        u, ts = engine.osc_law(n, self._params(ref_frame, xyz_offset), J, M, dq2, t2, g=g, Cdq=Cdq, xyz=xyz, R=R,
                               q=q2, target_velocity=tv2, integrated_error=ie, u_null_ext=une,
                               training_signal=True, device=device)
        if self.ki != 0:
            self.integrated_error = ie[0] if single else ie
        self.training_signal = ts[0] if single else ts
        return u[0] if single else u
