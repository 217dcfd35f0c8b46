"""The `robot_config` surface of abr_control (abr_control/arms/base_config.py:19-837),
evaluated by HIP kernels for batches of joint states.

Same names, arguments and return dtypes as the reference so controllers and example
scripts are drop-in:

    J/M/g/C/dJ/R -> float32 (the reference casts at base_config.py:223,247,270,285,301,336)
    Tx, T, T_inv, quaternion -> float64

`q` of shape (n,) returns exactly the reference's shapes; `q` of shape (B, n) returns the
same stacked along a leading batch axis.  `DeviceArray` inputs stay on the GPU and return
`DeviceArray`s in the kernel dtype (no float32 rounding).

There is no symbolic code generation and no cache directory: an arm is a small constant
table (tools/extract_arm_table.py derives it from any abr_control-style config).
"""
import ctypes as C

import numpy as np

from .. import _abi, engine
from .._lib import DeviceArray, check, lib


class BatchedConfig:
    """Base class of the arm configs.

    Parameters
    ----------
    table : dict
        arm table (see include/abrk.h `abrk_arm_desc`, tools/extract_arm_table.py)
    builtin : str or None
        name of a built-in arm -> compile-time specialised kernels; None -> the table is
        registered as a user arm (generic kernels)
    use_cython : ignored (accepted for drop-in compatibility, base_config.py:78)
    dtype : numpy float64 (default) or float32 - arithmetic of the kernels
    device : HIP device ordinal
    reference_dtypes : bool
        True (default): NumPy results carry the reference's dtypes (float32 for
        J/M/g/C/dJ/R).  False: keep the kernel dtype.
    compiled : None, True or False (user arms only)
        kernels specialised for this table (abr_control_amd/specialize.py - the counterpart of the reference's cached
        generated functions, base_config.py:173-191).  None (default): use the cached plugin when there is one, else
        the runtime-table kernels.  True: build it now if it is not cached (one hipcc run, 1-3 min).  False: never.
    """

    def __init__(self, table, builtin=None, use_cython=True, dtype=np.float64, device=0,
                 reference_dtypes=True, compiled=None, **kwargs):
        if kwargs:
            raise TypeError(f"unexpected keyword arguments {sorted(kwargs)}")  # as base_config.py:78 would
        self.table = table
        self.N_JOINTS = int(table["n_joints"])
        self.N_LINKS = int(table["n_links_dyn"])
        self.ROBOT_NAME = table.get("name", "robot")
        self.use_cython = use_cython
        self.dtype = np.dtype(dtype)
        self.device = device
        self.reference_dtypes = reference_dtypes
        self.x_zeros = np.zeros(3)
        n = self.N_JOINTS
        # reference attributes consumers read (interfaces/pygame.py:72-73, arm_sim.py:27-30)
        md = table["mdiag"]
        self._M_LINKS = [np.diag(np.asarray(md[l], dtype=float)) for l in range(len(md))]
        self._M_JOINTS = [np.zeros((6, 6)) for _ in range(n)]
        rows = [np.asarray(table["A0"])[:, 3]]
        for i in range(n):
            rows.append(np.asarray(table["AJ"][i])[:, 3])
            rows.append(np.asarray(table["B"][i])[:, 3])
        if table["has_ee"]:
            rows.append(np.asarray(table["E"])[:, 3])
        self.L = np.array(rows)
        if "START_ANGLES" in table:
            self.START_ANGLES = np.array(table["START_ANGLES"])
        self._builtin = builtin
        self._compiled = compiled
        self._plugin_path = None
        self._arm_id = None

    # ---- library handle (lazy: constructing a config needs no GPU)
    @property
    def arm_id(self):
        if self._arm_id is None:
            if self._builtin is not None:
                self._arm_id = check(lib().abrk_arm_builtin(self._builtin.encode()))
            else:
                desc = _abi.desc_from_table(self.table)
                path = None
                if self._compiled is not False:
                    from .. import specialize

                    path = specialize.compile_arm(self.table) if self._compiled else specialize.find_compiled(self.table)
                if path:
                    self._arm_id = check(lib().abrk_arm_create_compiled(C.byref(desc), path.encode()))
                    self._plugin_path = path
                else:
                    self._arm_id = check(lib().abrk_arm_create(C.byref(desc)))
        return self._arm_id

    @property
    def plugin_path(self):
        """the shared object of this arm's compiled kernels, or None (built-in arms; user arms on the runtime table)"""
        self.arm_id  # noqa: B018 - registers the arm, which is when the choice is made
        return self._plugin_path

    def compile(self, **kw):
        """build (or find) the kernels specialised for this arm's table and switch to them; returns the plugin path"""
        if self._builtin is not None:
            return None
        from .. import specialize

        path = specialize.compile_arm(self.table, **kw)
        self.close()
        self._compiled = True
        return path

    def close(self):
        """hand a user arm's registry slot back to the library (built-in arms hold none).  Called on garbage
        collection too, so loops that build configs repeatedly do not run out of the 4091 user slots."""
        arm_id, self._arm_id, self._plugin_path = self._arm_id, None, None
        if arm_id is not None and self._builtin is None:
            try:
                lib().abrk_arm_destroy(arm_id)
            except Exception:  # noqa: BLE001 - interpreter shutdown: the library may already be gone
                pass

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    # A copy (copy.copy / deepcopy / pickle) must not share the registry handle: the first of the two to be collected
    # would hand the slot back under the other.  Copies start unregistered and take a handle of their own at first use.
    # (The library also tags user-arm ids with a generation, so a stale id fails with ENOARM instead of reaching
    # whatever arm took the slot next.)
    def __getstate__(self):
        st = dict(self.__dict__)
        st["_arm_id"] = None
        st["_plugin_path"] = None
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)

    def __copy__(self):
        new = self.__class__.__new__(self.__class__)
        new.__setstate__(self.__getstate__())
        return new

    def __deepcopy__(self, memo):
        import copy

        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        new.__setstate__(copy.deepcopy(self.__getstate__(), memo))
        return new

    def frame_id(self, name):
        return _abi.frame_id(name, self.N_JOINTS)

    # ---- plumbing
    def _prep(self, q, dq=None):
        """-> (q2d, dq2d, single, on_device)"""
        if isinstance(q, DeviceArray):
            return q, dq, False, True
        q = np.asarray(q, dtype=self.dtype)
        single = q.ndim == 1
        q = np.atleast_2d(q)
        if q.shape[-1] != self.N_JOINTS:
            raise ValueError(f"q has {q.shape[-1]} joints, {self.ROBOT_NAME} has {self.N_JOINTS}")
        if dq is not None:
            dq = np.atleast_2d(np.asarray(dq, dtype=self.dtype))
        return q, dq, single, False

    def _eval(self, what, q, dq=None, name="EE", x=None, cast32=False):
        q2, dq2, single, dev = self._prep(q, dq)
        frame = self.frame_id(name)
        xo = None
        if x is not None and not np.allclose(x, 0):
            xo = np.asarray(x, dtype=float)
        res = engine.dynamics(self.arm_id, self.N_JOINTS, q2, dq2, frame, xo, (what,), self.dtype, self.device)[what]
        if dev:
            return res
        if cast32 and self.reference_dtypes:
            res = res.astype(np.float32)
        elif not cast32 and self.reference_dtypes:
            res = res.astype(np.float64)
        return res[0] if single else res

    def dynamics(self, q, dq=None, name="EE", x=None, want=("Tx", "J", "M", "g")):
        """Batched extension: several quantities from ONE kernel launch (shared forward
        kinematics); returns a dict of full-precision arrays."""
        q2, dq2, single, dev = self._prep(q, dq)
        xo = None if x is None or np.allclose(x, 0) else np.asarray(x, dtype=float)
        res = engine.dynamics(self.arm_id, self.N_JOINTS, q2, dq2, self.frame_id(name), xo, tuple(want), self.dtype,
                              self.device)
        if single and not dev:
            res = {k: v[0] for k, v in res.items()}
        return res

    # ---- the reference's wrappers (base_config.py:210-415)
    def g(self, q):
        """force of gravity in joint space (base_config.py:210-223)"""
        return self._eval("g", q, cast32=True)

    def dJ(self, name, q, dq, x=None):
        """derivative of the Jacobian wrt time (base_config.py:225-247)"""
        return self._eval("dJ", q, dq, name, x, cast32=True)

    def J(self, name, q, x=None):
        """Jacobian of point x in frame `name` (base_config.py:249-270)"""
        return self._eval("J", q, None, name, x, cast32=True)

    def M(self, q):
        """joint space inertia matrix (base_config.py:272-285)"""
        return self._eval("M", q, cast32=True)

    def R(self, name, q):
        """rotation matrix of frame `name` (base_config.py:287-301)"""
        return self._eval("R", q, None, name, cast32=True)

    def quaternion(self, name, q):
        """orientation of frame `name` as unit quaternion (w,x,y,z), w >= 0
        (base_config.py:304-318; computed from the full-precision rotation)"""
        return self._eval("quat", q, None, name)

    def C(self, q, dq):
        """Coriolis/centrifugal matrix such that C @ dq is the full term (base_config.py:320-336)"""
        return self._eval("C", q, dq, cast32=True)

    def T(self, name, q):
        """4x4 transform of frame `name` (base_config.py:338-369)"""
        return self._eval("T", q, None, name)

    def Tx(self, name, q, x=None):
        """world position of point x of frame `name` (base_config.py:371-392)"""
        return self._eval("Tx", q, None, name, x)

    def T_inv(self, name, q, x=None):
        """inverse transform of frame `name` (base_config.py:394-415)"""
        return self._eval("Tinv", q, None, name)
