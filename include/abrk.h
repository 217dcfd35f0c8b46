/*
 * abrk.h - C ABI of libabrk.so, the MI355X-native batched operational-space-control engine.
 *
 * This is the drop-in boundary for ONE hot path of abr/abr_control: the per-timestep
 * evaluation of an arm's kinematics/dynamics and the control laws built on it.  Every
 * entry point below names the reference interface (file:line under /root/reference)
 * it replaces.  The reference evaluates one joint state per call through SymPy-generated
 * Cython functions; this library evaluates a batch of B independent joint states per
 * call with hand-written HIP kernels for gfx950.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, caller-owned buffers, no allocation inside
 *     the compute calls.  Arrays are row-major [B, ...] exactly like the reference's
 *     per-call outputs stacked along a leading batch axis.
 *   - dtype selects the arithmetic AND the element type of every array argument:
 *     ABRK_F64 (double) or ABRK_F32 (float).
 *   - Every array pointer may be a device pointer (zero-copy) or a host pointer (the
 *     library stages it through device scratch on the given stream and copies results
 *     back before returning).  Detection is by hipPointerGetAttributes.
 *   - `device` is a HIP device ordinal; `stream` is a hipStream_t (NULL = default
 *     stream of that device).  Calls with device pointers are asynchronous on `stream`;
 *     calls that had to stage host memory synchronise the stream before returning.
 *   - Return value: 0 on success, negative ABRK_E* code on failure; abrk_last_error()
 *     returns a thread-local message.  There is NO CPU fallback: without a usable HIP
 *     device every compute call fails with ABRK_ENODEV.
 *   - Re-entrant per (device, stream): concurrent calls must use distinct streams.
 *
 * Frames are addressed by integer id: link_i -> 2*i, joint_i -> 2*i+1, "EE" ->
 * 2*n_joints+1 (the reference addresses them by the strings "link{i}", "joint{i}",
 * "EE": e.g. abr_control/arms/ur5/config.py:301-339).
 */
#ifndef ABRK_H
#define ABRK_H

#include <stddef.h>
#include <stdint.h>

#include "abrk_types.h" /* ABRK_VERSION, limits, dtype / error codes, abrk_arm_desc */

#ifdef __cplusplus
extern "C" {
#endif

/* Built-in arms ("ur5", "jaco2", "twojoint", "threejoint", "onejoint"): compile-time
 * specialised kernels.  Returns an arm id >= 0 or ABRK_ENOARM.                        */
int abrk_arm_builtin(const char* name);
/* Register a user arm (runtime table; generic kernels).  Returns arm id >= 0.
 * Replaces writing a BaseConfig subclass + first-use SymPy/Cython code generation
 * (base_config.py:125-146).                                                            */
int abrk_arm_create(const abrk_arm_desc* desc);
/* Register a user arm together with kernels compiled for its own table - the counterpart of the reference's cached
 * generated functions (base_config.py:173-191: `_load_from_file` imports the Cython module generated for this arm;
 * _generate_and_save_function :125-146 writes it).  `plugin_path` is a shared object built by
 * `make -C abr_control_amd/csrc plugin` from the same description (abr_control_amd/specialize.py generates the source,
 * runs the build and caches the result); it is checked against `desc` value by value and against abrk_plugin_abi(), and
 * refused with ABRK_EINVAL on any mismatch.  The arm then runs the same compile-time specialised kernels as a built-in
 * arm (about 2.2x the rate of the runtime-table kernels abrk_arm_create gives).  Returns arm id >= 0.                */
int abrk_arm_create_compiled(const abrk_arm_desc* desc, const char* plugin_path);
/* tag of the kernel headers and compile flags this library was built from; a plugin must carry the same one */
const char* abrk_plugin_abi(void);
int abrk_arm_get_desc(int arm_id, abrk_arm_desc* out);
int abrk_arm_destroy(int arm_id);

/* ---------------------------------------------------------------------------------
 * Dynamics: replaces the generated `autofunc_c(q0..,[dq0..],[x,y,z])` calls behind
 * BaseConfig.Tx/J/M/g/C/dJ/R/T/T_inv (base_config.py:210-415).  One launch evaluates
 * every requested quantity for all B states; FK is shared between them.
 *   want-mask bit      output (row-major per state)                reference wrapper
 *   ABRK_WANT_TX       Tx  [B,3]      position of x in `frame`     base_config.py:371
 *   ABRK_WANT_J        J   [B,6,n]    Jacobian of that point       base_config.py:249
 *   ABRK_WANT_M        M   [B,n,n]    joint-space inertia          base_config.py:272
 *   ABRK_WANT_G        g   [B,n]      gravity torque               base_config.py:210
 *   ABRK_WANT_C        C   [B,n,n]    Christoffel Coriolis matrix  base_config.py:320
 *   ABRK_WANT_DJ       dJ  [B,6,n]    time derivative of J         base_config.py:225
 *   ABRK_WANT_R        R   [B,3,3]    rotation of `frame`          base_config.py:287
 *   ABRK_WANT_T        T   [B,4,4]    transform of `frame`         base_config.py:338
 *   ABRK_WANT_TINV     Ti  [B,4,4]    inverse transform            base_config.py:394
 *   ABRK_WANT_QUAT     quat[B,4]      (w,x,y,z) of `frame`         base_config.py:304
 * Outputs keep full `dtype` precision; the reference's float32 rounding of
 * J/M/g/C/dJ/R (base_config.py:223,247,270,285,301,336) is applied by the Python layer.
 * q: [B,n]; dq: [B,n] (needed for C/dJ, else may be NULL); x_off: [3] host values or NULL.
 * --------------------------------------------------------------------------------- */
enum {
  ABRK_WANT_TX = 1u << 0,
  ABRK_WANT_J = 1u << 1,
  ABRK_WANT_M = 1u << 2,
  ABRK_WANT_G = 1u << 3,
  ABRK_WANT_C = 1u << 4,
  ABRK_WANT_DJ = 1u << 5,
  ABRK_WANT_R = 1u << 6,
  ABRK_WANT_T = 1u << 7,
  ABRK_WANT_TINV = 1u << 8,
  ABRK_WANT_QUAT = 1u << 9
};

typedef struct abrk_dyn_out {
  void* Tx;
  void* J;
  void* M;
  void* g;
  void* C;
  void* dJ;
  void* R;
  void* T;
  void* Tinv;
  void* quat;
} abrk_dyn_out;

int abrk_dynamics_batch(int arm_id, int dtype, int64_t B, const void* q, const void* dq,
                        int frame, const double* x_off, uint32_t want,
                        const abrk_dyn_out* out, int device, void* stream);

/* ---------------------------------------------------------------------------------
 * Secondary (null-space / joint-space) controllers: Damping.generate
 * (controllers/damping.py:21-32), RestingConfig.generate (resting_config.py:18-42) and
 * Joint.generate (joint.py:104-131, angle states only).
 * --------------------------------------------------------------------------------- */
enum { ABRK_NULL_DAMPING = 1, ABRK_NULL_RESTING = 2 };

typedef struct abrk_null_ctrl {
  int32_t kind;
  int32_t rest_mask[ABRK_MAX_JOINTS]; /* RestingConfig.rest_indices                  */
  double kp;                          /* Joint kp (resting)                          */
  double kv;                          /* Damping kv / Joint kv                       */
  double rest_angles[ABRK_MAX_JOINTS];
} abrk_null_ctrl;

/* OSC constructor arguments (controllers/osc.py:53-118), passed through unchanged; the
 * derived constants (task_space_gains, lamb, sat_gain_*, scale_*) are formed inside
 * with the reference's expressions.                                                    */
typedef struct abrk_osc_params {
  double kp, ko, kv, ki;
  int32_t use_vmax;
  int32_t use_g;
  int32_t use_C;
  int32_t orientation_algorithm;
  double vmax[2];
  int32_t ctrlr_dof[6];
  int32_t ref_frame;            /* frame id of generate(ref_frame=...), osc.py:218       */
  int32_t n_null;
  double xyz_offset[3];         /* generate(xyz_offset=...), zeros == None               */
  abrk_null_ctrl null_ctrl[ABRK_MAX_NULL];
} abrk_osc_params;

/* OSC.generate (controllers/osc.py:217-320) for B independent states.
 *   q, dq [B,n]; target [B,6]; target_velocity [B,6] or NULL (== zeros, osc.py:239);
 *   integrated_error [B,6] in/out state, required iff ki != 0 (osc.py:81,262-264);
 *   u_null_ext [B,n] or NULL: extra secondary control signal already evaluated by the
 *     caller (any Python null controller), projected into the null space with the same
 *     filter as osc.py:315-318;
 *   u [B,n] out; training_signal [B,n] out or NULL (osc.py:297).
 * training_signal == NULL changes the LAST BITS of u on one family of laws: the six-row law (any ctrlr_dof beyond x,y,z
 * of the end effector) without optional inputs (no target_velocity / integrated_error / u_null_ext / null_ctrl) then
 * runs kernels that never form the training signal (`osc_kernel<.., NOTS = true>`: the gravity term joins the velocity
 * term before the factorisations instead of being subtracted from the finished sum).  Same law, same accuracy against
 * the reference (both forms are held to the same 1e-6 against its outputs: tests/test_gpu_parity.py
 * test_gpu_six_row_cases_without_training_signal), a different rounding order: u differs by ~1e-16 of its scale times
 * cond(Mx_inv).  For a given choice of NULL / non-NULL a row's bits depend on nothing else (batch size, position in the
 * batch, sharding).  Every other law returns the same bits either way.                                             */
int abrk_osc_generate_batch(int arm_id, int dtype, const abrk_osc_params* params, int64_t B,
                            const void* q, const void* dq, const void* target,
                            const void* target_velocity, void* integrated_error,
                            const void* u_null_ext, void* u, void* training_signal,
                            int device, void* stream);

/* OSC.generate AND the robot_config outputs it consumed, in one launch (one forward kinematics): besides u, any of
 *   Tx [B,3] = robot_config.Tx(ref_frame, q, x=xyz_offset)   (base_config.py:371-392)
 *   J  [B,6,n] = robot_config.J(ref_frame, q, x=xyz_offset)  (:249-270)
 *   M  [B,n,n] = robot_config.M(q)                            (:272-285)
 *   g  [B,n]   = robot_config.g(q)                            (:210-223)
 *   C  [B,n,n] = robot_config.C(q, dq)                        (:320-336)   } the velocity-dependent functions:
 *   dJ [B,6,n] = robot_config.dJ(ref_frame, q, dq, x=xyz_offset) (:225-247) } +288 B per UR5 row each
 * selected by `want` (ABRK_WANT_TX | ABRK_WANT_J | ABRK_WANT_M | ABRK_WANT_G | ABRK_WANT_C | ABRK_WANT_DJ; the other
 * fields of `out` are ignored), in the kernel's arithmetic type.  For callers that read those next to ctrlr.generate - adaptive terms on
 * training_signal, logging, a second controller on the same state (the consumers of osc.py:242-301) - and would
 * otherwise evaluate the kinematics twice.  840 B of algorithmic traffic per UR5 row in fp64: the HBM-bound form of
 * the path (SURVEY.md 8d "Mode F").  Everything else as abrk_osc_generate_batch.                              */
int abrk_osc_generate_full_batch(int arm_id, int dtype, const abrk_osc_params* params, int64_t B,
                                 const void* q, const void* dq, const void* target,
                                 const void* target_velocity, void* integrated_error,
                                 const void* u_null_ext, void* u, void* training_signal, uint32_t want,
                                 const abrk_dyn_out* out, int device, void* stream);

/* The WAVE-COOPERATIVE mapping of the same law (K = lanes_per_arm in {4, 8, 16} lanes per arm instance, frames /
 * Jacobian columns / M staged in LDS, abr_control_amd/csrc/abrk_coop.h): a measurement variant kept beside the
 * lane-per-arm kernels for the config-sized batch (profiles/round2/coop_ab.md).  Built-in ur5, fp64, the plain law of
 * BASELINE config 2 (x,y,z of the EE; no xyz_offset, null controllers, Coriolis term or integral term).      */
int abrk_osc_generate_coop_batch(int arm_id, int dtype, const abrk_osc_params* params, int64_t B, const void* q,
                                 const void* dq, const void* target, void* u, void* training_signal,
                                 int lanes_per_arm, int device, void* stream);

/* The same call over SEVERAL devices (BASELINE config 4: 2^20 rows over the 8 GPUs of a node): host arrays in, host
 * arrays out; the batch is cut into n_shards contiguous row ranges (sizes differing by at most one row), shard g is
 * evaluated on devices[g] (a device may appear more than once), each on a stream of its own, all kernels in flight before the first result is collected.  Rows are
 * independent: no exchange step, no collective.  Per-row state (integrated_error) stays with its shard.            */
int abrk_osc_generate_sharded(int arm_id, int dtype, const abrk_osc_params* params, int64_t B, const void* q,
                              const void* dq, const void* target, const void* target_velocity,
                              void* integrated_error, const void* u_null_ext, void* u, void* training_signal,
                              int n_shards, const int* devices);

/* The same control law (osc.py:244-318) on CALLER-SUPPLIED dynamics: keeps `OSC(robot_config=<any duck
 * type>)` working for configs whose arithmetic lives elsewhere (the reference's MujocoConfig,
 * abr_control/arms/mujoco_config.py:201-451: J/M/g/Tx/R read from mjData).  Per row, row-major:
 *   J [B,6,n] = robot_config.J(ref_frame,q,x); M [B,n,n] = robot_config.M(q);
 *   g [B,n] (iff use_g); Cdq [B,n] = C(q,dq)@dq (iff use_C); xyz [B,3] = robot_config.Tx(...) (iff any
 *   of ctrlr_dof[0:3]); R [B,3,3] = robot_config.R(...) (iff any of ctrlr_dof[3:6]);
 *   q [B,n] only for a fused RestingConfig; the remaining arguments as abrk_osc_generate_batch.
 * params->ref_frame / xyz_offset are ignored (already applied by whoever produced J, xyz, R).    */
int abrk_osc_law_batch(int n_joints, int dtype, const abrk_osc_params* params, int64_t B, const void* J,
                       const void* M, const void* g, const void* Cdq, const void* xyz, const void* R,
                       const void* q, const void* dq, const void* target, const void* target_velocity,
                       void* integrated_error, const void* u_null_ext, void* u, void* training_signal,
                       int device, void* stream);

/* The helper methods of OSC that callers (and the reference's own tests, controllers/tests/test_osc.py:12-140)
 * use directly.  The fused kernels above carry the same steps inline; these are their general forms.
 *
 * OSC._Mx (controllers/osc.py:120-147): task-space inertia Mx = (J M^-1 J^T)^-1, inverse while
 * |det| >= threshold, else pinv(rcond = 0.1 threshold).
 *   M [B,n,n] symmetric positive definite; J [B,k,n] = the k task rows OSC keeps (J[ctrlr_dof]), 1 <= k <= 6;
 *   Mx [B,k,k] out; M_inv [B,n,n] out or NULL.                                                    */
int abrk_osc_mx_batch(int n_joints, int k, int dtype, int64_t B, const void* M, const void* J, double threshold,
                      void* Mx, void* M_inv, int device, void* stream);
/* OSC._velocity_limiting (osc.py:198-215): u_task [B,6] -> out [B,6] with kp, ko, kv, vmax of params
 * (params->use_vmax must be set).                                                                 */
int abrk_osc_velocity_limiting_batch(int dtype, const abrk_osc_params* params, int64_t B, const void* u_task,
                                     void* out, int device, void* stream);
/* OSC._calc_orientation_forces (osc.py:149-196) from R [B,3,3] = robot_config.R(ref_frame, q) (ABRK_WANT_R of
 * abrk_dynamics_batch) and target_abg [B,3] (Euler angles, 'rxyz'); algorithm 0 or 1; out [B,3]. */
int abrk_osc_orientation_forces_batch(int algorithm, int dtype, int64_t B, const void* R, const void* target_abg,
                                      void* u_task_orientation, int device, void* stream);

/* The functions of abr_control/utils/transformations.py that the control path uses, B rows per call:
 *   ABRK_TF_QUAT_FROM_EULER_RXYZ / _SXYZ  a [B,3] angles          -> out [B,4] (w,x,y,z)   transformations.py:1096
 *   ABRK_TF_QUAT_FROM_MATRIX              a [B,3,3] rotations     -> out [B,4], w >= 0     :1192
 *   ABRK_TF_QUAT_MULTIPLY                 a = q1, b = q0 [B,4]    -> out [B,4]             :1274
 *   ABRK_TF_QUAT_CONJUGATE                a [B,4]                 -> out [B,4]             :1293
 *   ABRK_TF_UNIT_VECTOR4 / _VECTOR3       a [B,4] / [B,3]         -> out same shape        :1632
 *   ABRK_TF_EULER_MATRIX_RXYZ             a [B,3] angles          -> out [B,3,3]           :973
 * b is NULL except for the product.                                                               */
enum {
  ABRK_TF_QUAT_FROM_EULER_RXYZ = 0, ABRK_TF_QUAT_FROM_EULER_SXYZ = 1, ABRK_TF_QUAT_FROM_MATRIX = 2,
  ABRK_TF_QUAT_MULTIPLY = 3, ABRK_TF_QUAT_CONJUGATE = 4, ABRK_TF_UNIT_VECTOR4 = 5, ABRK_TF_UNIT_VECTOR3 = 6,
  ABRK_TF_EULER_MATRIX_RXYZ = 7
};
int abrk_transformations_batch(int op, int dtype, int64_t B, const void* a, const void* b, void* out, int device,
                               void* stream);

/* Launch plans for control loops that call the same law on the same device buffers every tick (the
 * shape of every example loop, examples/PyGame/force_osc_xy.py:57-78): all arguments of
 * abrk_osc_generate_batch are validated and converted ONCE; abrk_plan_launch then only enqueues the
 * kernel on the plan's stream (no pointer classification, no locks, no staging).  Every array must be a
 * device pointer.  Returns a plan id >= 0.                                                       */
int abrk_osc_plan_create(int arm_id, int dtype, const abrk_osc_params* params, int64_t B, const void* q,
                         const void* dq, const void* target, const void* target_velocity,
                         void* integrated_error, const void* u_null_ext, void* u, void* training_signal,
                         int device, void* stream);
/* General form: RECORD one control tick.  Between abrk_plan_begin and abrk_plan_end every abrk_*_batch call made by
 * this thread (dynamics, OSC, OSC law, Sliding, Joint / Damping / RestingConfig, AvoidJointLimits, Floating,
 * AvoidObstacles, inverse kinematics, the two-link plant step and rollout) is validated and converted as usual but
 * its kernel launch is kept in the plan instead of being enqueued - e.g. AvoidJointLimits and AvoidObstacles
 * accumulating into the buffer that the OSC call then takes as u_null_ext (osc.py:310-318) become ONE plan of
 * three kernels.  Each recorded call must pass device pointers only and the device and stream given to
 * abrk_plan_begin.  abrk_plan_end returns the plan id (>= 0); abrk_plan_abort drops the recording.  Ids of
 * destroyed plans are never valid again (generation-tagged), their slots and memory are recycled: a loop that
 * re-plans whenever gains, targets or buffers change can do so indefinitely; up to 4096 plans live at a time. */
int abrk_plan_begin(int device, void* stream);
int abrk_plan_end(void);
int abrk_plan_abort(void);
int abrk_plan_count(void); /* live plans */
int abrk_plan_launch(int plan);
/* `repeat` consecutive plain launches of the plan enqueued by ONE call (no per-launch crossing of the language
 * boundary): for a few tens of ticks this beats the fixed cost of a graph launch.                              */
int abrk_plan_launch_repeat(int plan, int repeat);
/* `repeat` consecutive launches of the plan as ONE hipGraph launch (captured on first use and cached per
 * repeat count): removes the per-launch host work and tightens the dependent-launch gaps of short kernels. */
int abrk_plan_launch_graph(int plan, int repeat);
int abrk_plan_destroy(int plan);

/* Sliding.generate (controllers/sliding.py:34-99), cartesian=True or False.
 *   target [B,3] (cartesian) or [B,n]; target_velocity / target_acc same shape or NULL
 *   (== 0); u [B,n] out; s [B,n] out or NULL (Sliding.s, sliding.py:89).              */
typedef struct abrk_sliding_params {
  double kd, lamb;
  int32_t cartesian;
  int32_t ref_frame;
  double offset[3];
} abrk_sliding_params;

int abrk_sliding_generate_batch(int arm_id, int dtype, const abrk_sliding_params* params,
                                int64_t B, const void* q, const void* dq, const void* target,
                                const void* target_velocity, const void* target_acc, void* u,
                                void* s, int device, void* stream);
/* Launch plan for Sliding.generate on fixed device buffers (see abrk_osc_plan_create; launched, replayed as a
 * hipGraph and destroyed by abrk_plan_launch / abrk_plan_launch_graph / abrk_plan_destroy).      */
int abrk_sliding_plan_create(int arm_id, int dtype, const abrk_sliding_params* params, int64_t B, const void* q,
                             const void* dq, const void* target, const void* target_velocity, const void* target_acc,
                             void* u, void* s, int device, void* stream);

/* ---------------------------------------------------------------------------------
 * Closed loop on the device: the reference's examples alternate OSC.generate and
 * the two-link plant step ArmSim._step (abr_control/arms/twojoint/arm_sim.py:101-137) once per
 * millisecond of simulated time (examples/PyGame/force_osc_xy.py:57-78).  K1..K4 are the constants
 * ArmSim.__init__ derives (arm_sim.py:26-41); dt the Euler step.
 * --------------------------------------------------------------------------------- */
typedef struct abrk_twolink_plant {
  double K1, K2, K3, K4;
  double dt;
} abrk_twolink_plant;

/* ArmSim._step for B arms: (q, dq) [B,2] advanced in place by the torques u [B,2].            */
int abrk_twolink_step_batch(int dtype, const abrk_twolink_plant* plant, int64_t B, void* q, void* dq,
                            const void* u, int device, void* stream);

/* n_steps x { u = OSC.generate(q, dq, target); ArmSim._step(u) } in ONE launch, state in registers.
 *   arm_id: a two-joint arm; q, dq [B,2] in/out; target [B,6]; integrated_error [B,6] in/out (ki != 0);
 *   checkpoints (optional, every `every` steps, n_chk = n_steps / every):
 *   q_traj, dq_traj, u_traj [B, n_chk, 2] or NULL.                                              */
int abrk_osc_rollout_twolink_batch(int arm_id, int dtype, const abrk_osc_params* params,
                                   const abrk_twolink_plant* plant, int64_t B, int32_t n_steps, int32_t every,
                                   void* q, void* dq, const void* target, void* integrated_error,
                                   void* q_traj, void* dq_traj, void* u_traj, int device, void* stream);

/* ---------------------------------------------------------------------------------
 * Iterative inverse kinematics: InverseKinematics.generate_path
 * (abr_control/controllers/path_planners/inverse_kinematics.py:28-135) for B independent paths, all
 * n_timesteps iterations of a path inside one kernel (each iteration: Tx, J, quaternion of the EE, two
 * pseudo-inverses).  max_dx / max_dr / max_dq are the constructor values (per second); they are scaled by
 * dt exactly as generate_path does.  method 1: resolved motion, 2: damped least squares, 3: position with
 * orientation in its null space (the reference default).  target [B,6] = xyz + Euler angles 'sxyz'.
 * position [B,n] start joint angles; position_path, velocity_path [B, n_timesteps, n] out.
 * --------------------------------------------------------------------------------- */
typedef struct abrk_ik_params {
  double max_dx, max_dr, max_dq, dt;
  int32_t n_timesteps;
  int32_t method;
} abrk_ik_params;

int abrk_ik_generate_path_batch(int arm_id, int dtype, const abrk_ik_params* params, int64_t B,
                                const void* position, const void* target, void* position_path,
                                void* velocity_path, int device, void* stream);

/* Joint.generate (controllers/joint.py:104-131) / Damping / RestingConfig standalone.
 *   ctrl.kind == ABRK_NULL_DAMPING: u = M (-kv dq)            (damping.py:31-32)
 *   ctrl.kind == ABRK_NULL_RESTING: RestingConfig.generate     (resting_config.py:33-42)
 *   ctrl.kind == 0: Joint.generate with target [B,n], target_velocity [B,n] or NULL,
 *                   account_for_gravity as given.                                       */
int abrk_joint_generate_batch(int arm_id, int dtype, const abrk_null_ctrl* ctrl,
                              int account_for_gravity, int64_t B, const void* q,
                              const void* dq, const void* target, const void* target_velocity,
                              void* u, int device, void* stream);

/* The same for Sliding.generate (sliding.py:34-99; BASELINE config 5), Joint / Damping / RestingConfig.generate
 * (joint.py:104-131, damping.py:21-32, resting_config.py:18-42) and the robot_config functions (base_config.py:210-415):
 * arguments as the *_batch entry point of the same name, host arrays only, n_shards contiguous row ranges on
 * devices[0..n_shards-1], no collective.  The shards of one device are staged in, launched and collected by one host
 * thread per device (the caller's own for the first device named), under that device's lock only: staging copies of
 * different devices overlap, and calls on disjoint device sets do not wait for each other.  Shards that should STAY
 * on the devices: the *_resident entry points below.                                                              */
int abrk_sliding_generate_sharded(int arm_id, int dtype, const abrk_sliding_params* params, int64_t B, const void* q,
                                  const void* dq, const void* target, const void* target_velocity,
                                  const void* target_acc, void* u, void* s_out, int n_shards, const int* devices);
int abrk_joint_generate_sharded(int arm_id, int dtype, const abrk_null_ctrl* ctrl, int account_for_gravity, int64_t B,
                                const void* q, const void* dq, const void* target, const void* target_velocity, void* u,
                                int n_shards, const int* devices);
int abrk_dynamics_sharded(int arm_id, int dtype, int64_t B, const void* q, const void* dq, int frame,
                          const double* x_off, uint32_t want, const abrk_dyn_out* out, int n_shards,
                          const int* devices);

/* ---- RESIDENT SHARDS ("results remain in per-device buffers unless the caller asks for host arrays";
 * north_star: "the batch shards trivially across the 8 GPUs of one node").  A batch whose shards LIVE on the devices, driven
 * by ONE host thread: every array argument is a table of n_shards DEVICE pointers - shard g holds rows[g] rows of every
 * array on devices[g] (a device may appear more than once; a NULL table = the array is absent).  A call only ENQUEUES:
 * shard g's kernels go to streams[g] (streams == NULL: the library's own stream of (device, k) for the k-th shard on
 * that device, abrk_shard_stream), nothing is staged, nothing is waited for; per-row state (integrated_error) stays
 * with its shard.  Semantics per shard are those of the *_batch entry point of the same name on (devices[g],
 * streams[g]) - the call IS that entry point, once per shard - so results are bit-identical to the unsharded call on
 * the same rows.  abrk_shards_sync drains every shard's stream and reports (once) ABRK_ESINGULAR of any of them.
 * Control loops record one plan per shard (abrk_plan_begin(devices[g], streams[g]) ... abrk_plan_end) and replay all of
 * them with one abrk_plans_launch per tick / per K ticks.  No collective, no exchange step: rows are independent.
 * Replaces: the reference evaluates one state per Python call (controllers/osc.py:217-320); its usage model - one
 * process owning the whole control loop, examples/PyGame/force_osc_xy.py:57-78 - scaled to N GPUs.               */
typedef struct abrk_shard_cut {
  int32_t n_shards;
  const int32_t* devices; /* [n_shards] */
  const int64_t* rows;    /* [n_shards]; 0 = the shard is skipped */
  void* const* streams;   /* [n_shards] or NULL */
} abrk_shard_cut;
void* abrk_shard_stream(int device, int slot); /* the stream a NULL `streams` stands for; NULL on failure */
int abrk_osc_generate_resident(int arm_id, int dtype, const abrk_osc_params* params, const abrk_shard_cut* cut,
                               const void* const* q, const void* const* dq, const void* const* target,
                               const void* const* target_velocity, void* const* integrated_error,
                               const void* const* u_null_ext, void* const* u, void* const* training_signal);
int abrk_sliding_generate_resident(int arm_id, int dtype, const abrk_sliding_params* params, const abrk_shard_cut* cut,
                                   const void* const* q, const void* const* dq, const void* const* target,
                                   const void* const* target_velocity, const void* const* target_acc, void* const* u,
                                   void* const* s_out);
int abrk_joint_generate_resident(int arm_id, int dtype, const abrk_null_ctrl* ctrl, int account_for_gravity,
                                 const abrk_shard_cut* cut, const void* const* q, const void* const* dq,
                                 const void* const* target, const void* const* target_velocity, void* const* u);
/* out: [n_shards] abrk_dyn_out, one per shard (only the fields selected by `want` are read) */
int abrk_dynamics_resident(int arm_id, int dtype, const abrk_shard_cut* cut, const void* const* q,
                           const void* const* dq, int frame, const double* x_off, uint32_t want,
                           const abrk_dyn_out* out);
int abrk_shards_sync(const abrk_shard_cut* cut);
/* `repeat` ticks of SEVERAL plans (one per shard) from one call: mode 0 = plain launches, 1 = one hipGraph of `repeat`
 * ticks per plan (abrk_plan_launch_graph).  Returns when every plan's work is enqueued.                          */
int abrk_plans_launch(const int* plans, int n_plans, int repeat, int mode);

/* ---------------------------------------------------------------------------------
 * The remaining secondary controllers.  Each writes u [B,n]; with accumulate != 0 it
 * adds to u instead (several secondary controllers summed on device before OSC's null-space filter,
 * abr_control/controllers/osc.py:310-318 - pass the sum as u_null_ext to abrk_osc_generate_batch).
 * --------------------------------------------------------------------------------- */

/* AvoidJointLimits (controllers/avoid_joint_limits.py:35-142).  The limit arrays are the ones the
 * reference's constructor stores (avoid_joint_limits.py:45-75): shifted by -pi, min/max swapped where
 * cross_zero, "no limit" (NaN there) flagged in no_limits_min / no_limits_max.                       */
typedef struct abrk_limits_params {
  double min_joint_angles[ABRK_MAX_JOINTS];
  double max_joint_angles[ABRK_MAX_JOINTS];
  double max_torque[ABRK_MAX_JOINTS];
  int32_t cross_zero[ABRK_MAX_JOINTS];
  int32_t gradient[ABRK_MAX_JOINTS];
  int32_t no_limits_min[ABRK_MAX_JOINTS];
  int32_t no_limits_max[ABRK_MAX_JOINTS];
} abrk_limits_params;

/* AvoidJointLimits.generate (avoid_joint_limits.py:83-142); depends on q only. */
int abrk_avoid_joint_limits_generate_batch(int n_joints, int dtype, const abrk_limits_params* params, int64_t B,
                                           const void* q, void* u, int accumulate, int device, void* stream);

/* Floating.generate (controllers/floating.py:27-71): gravity compensation in joint space or through
 * the task-space inertia of the EE (floating.py:42-62), minus M dq when dynamic (floating.py:66-69).
 * dq [B,n] is read only when dynamic != 0.                                                          */
int abrk_floating_generate_batch(int arm_id, int dtype, int dynamic, int task_space, int64_t B, const void* q,
                                 const void* dq, void* u, int accumulate, int device, void* stream);

/* AvoidObstacles.generate (controllers/avoid_obstacles.py:38-120): Khatib potential field between each
 * arm segment (joint_i .. joint_i+1 / EE) and each obstacle [x, y, z, radius], mapped through the
 * inertia of the closest point.  Obstacles are shared by all rows (set_obstacles, avoid_obstacles.py:122). */
#define ABRK_MAX_OBSTACLES 16
typedef struct abrk_obstacles_params {
  int32_t n_obstacles;
  int32_t reserved;
  double threshold, gain, maximum;
  double obstacles[ABRK_MAX_OBSTACLES][4];
} abrk_obstacles_params;

int abrk_avoid_obstacles_generate_batch(int arm_id, int dtype, const abrk_obstacles_params* params, int64_t B,
                                        const void* q, void* u, int accumulate, int device, void* stream);

/* ---------------------------------------------------------------------------------
 * Device plumbing (the host side is Python + ctypes; no PyTorch involved).
 * --------------------------------------------------------------------------------- */
int abrk_device_count(void);
int abrk_device_name(int device, char* buf, size_t len);
void* abrk_malloc(int device, size_t bytes);               /* NULL on failure */
int abrk_free(int device, void* p);
int abrk_memcpy_h2d(int device, void* dst, const void* src, size_t bytes, void* stream);
int abrk_memcpy_d2h(int device, void* dst, const void* src, size_t bytes, void* stream);
int abrk_memset(int device, void* dst, int value, size_t bytes, void* stream);
void* abrk_stream_create(int device);
int abrk_stream_destroy(int device, void* stream);
int abrk_stream_sync(int device, void* stream);
int abrk_device_sync(int device);
void* abrk_event_create(int device);
int abrk_event_destroy(int device, void* ev);
int abrk_event_record(int device, void* ev, void* stream);
int abrk_event_elapsed_ms(int device, void* start, void* stop, float* ms); /* syncs stop */

/* Diagnostics of the library's own device scratch (what the tests and a long-running host watch): the six-row OSC law
 * (any ctrlr_dof beyond x,y,z; osc.py:134-147 - its truncating pinv) parks the rows that need the eigen-decomposition in
 * a per-(device, stream) worklist; abrk_stream_destroy hands a stream's slot back.  Counters are process-wide and
 * monotonic. */
typedef struct abrk_scratch_info {
  int64_t worklist_slots;      /* (device, stream) pairs that currently hold six-row scratch, all devices */
  int64_t worklist_bytes;      /* device memory they hold */
  int64_t inline_fallbacks;    /* six-row calls that ran one-pass because no scratch could be had */
  int64_t evictions;           /* slots of idle / vanished streams that were recycled */
  int64_t device_free_bytes;   /* hipMemGetInfo of `device` */
  int64_t device_total_bytes;
  int64_t status_words_out;    /* ABRK_ESINGULAR words in use: one per live thread that made a host-array OSC call, one
                                  per (device, stream) that took a device-pointer OSC call and was not destroyed      */
  int64_t status_blocks;       /* pinned 4 KiB blocks (64 words each) the pool holds; never shrinks                  */
} abrk_scratch_info;
int abrk_scratch_stats(int device, abrk_scratch_info* out);

const char* abrk_last_error(void);
int abrk_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ABRK_H */
