/*
 * sccbf.h - C ABI of the batched control-barrier-function QP safety filter
 *           for AMD MI355X (gfx950).
 *
 * The reference (Safety-Critical-Control-WIRIN/sccav_cbf) is pure Python and
 * has no FFI seam; each entry point below names the reference code whose
 * arithmetic it replaces.  The Python classes that mirror the reference API
 * (sccav_cbf_amd.cbf.*) and the batched torch front-end
 * (sccav_cbf_amd.BatchedCBF) bind these symbols with ctypes.
 *
 * Conventions
 *   - every data pointer is a DEVICE pointer owned by the caller
 *     (hipMalloc / a torch tensor's data_ptr()); the library allocates nothing
 *   - every call is asynchronous on `stream` (a hipStream_t, NULL = default
 *     stream); the caller synchronises
 *   - return value 0 = launched, <0 = argument / launch error
 *     (sccbf_strerror); no exception crosses the ABI
 *   - calls are re-entrant and thread-safe; the only state the library keeps is one flag per device that
 *     says a one-time kernel attribute (more than 64 KB of LDS for long row lists) has been set
 *
 * Memory layout ("T" = float for *_f32, double for *_f64; A agents, M slots)
 *   state    T[A][4]        x, y, theta(yaw), v         one 4-vector / agent
 *   u_ref    T[A][2]        DBM: a_ref, delta_ref       KBM: v_ref, delta_ref
 *   obs      T[M][2][A][4]  obstacle slot m of agent a = the two 4-vectors
 *                           obs[m][0][a][:] and obs[m][1][a][:]  (8 scalars);
 *                           with SCCBF_FLAG_OBS_SHARED the table is T[M][2][4]
 *                           and is broadcast to every agent; with
 *                           SCCBF_FLAG_OBS_TILED it is T[ceil(A/64)][M][2][64][4],
 *                           agent a at tile a/64, position a%64 (the faster
 *                           layout: one contiguous block per 64 agents); with
 *                           SCCBF_FLAG_OBS_AGENT_MAJOR it is T[A][M][2][4] (the faster
 *                           layout for more than 128 slots)
 *   obs_type u8[M]          slot kind, shared by all agents; with
 *                           SCCBF_FLAG_TYPE_PER_AGENT it is u8[M][A]
 *   u_out    T[A][2]        DBM: a*, delta*             KBM: v*, delta*
 *   status   i32[A]         SCCBF_STATUS_*
 *   rows_out T[M][A][4]     optional (may be NULL): Lg_h[0], Lg_h[1], b, h of
 *                           every row  Lg_h . u >= b   before the QP
 *
 * Obstacle slot contents (8 scalars o[0..7]) by kind
 *   SCCBF_OBS_ELLIPSE  cx, cy, a, b, theta_o, vx, vy, -      (a, b include buffer)
 *   SCCBF_OBS_CONE     cx, cy, theta_o, v_o, a, beta, -, -   (a includes buffer)
 *   SCCBF_OBS_LANE     c0, c1, c2, c3, buffer, -, -, -       y = c0+c1 x+c2 x^2+c3 x^3
 *   SCCBF_OBS_LANE6    c0, c1, c2, c3, c4, c5, c6, buffer    the same barrier for a polynomial of degree up to 6
 *   SCCBF_OBS_RADIAL   cx, cy, a, b, cdot_x, cdot_y, kv, -
 *   SCCBF_OBS_EMPTY    ignored (row 0.u >= -1, always satisfied)
 * Prepared kinds: the same obstacles with everything that does not depend on the agent already
 * evaluated by whoever writes the slot (sccbf_prepare_obs_*, sccbf_ingest_bbox_*), so that the
 * filter does no sine, cosine or reciprocal per slot and tick; same 8 scalars, same rows
 *   SCCBF_OBS_ELLIPSE_PREP  cx, cy, 1/a^2, 1/b^2, cos(theta_o), sin(theta_o), vx, vy
 *   SCCBF_OBS_CONE_PREP     cx, cy, v_o cos(theta_o + beta), v_o sin(theta_o + beta), a, beta, -, -
 * Box kinds: the slot is the obstacle's bounding box as perception delivers it (BoundingBox, reference
 * cbf/obstacles.py:59-64) plus the safety buffer, and the filter makes the obstacle from it on the fly,
 * as ObstacleList2D.update_by_bounding_box does on every tick (833-858): no ingestion pass in between
 *   SCCBF_OBS_ELLIPSE_BOX   loc x, loc y, extent x, extent y, yaw, speed, buffer, -
 *                           Ellipse2D.from_bounding_box (320-331): a = extent x + buffer,
 *                           b = extent y + buffer, theta_o = yaw, at rest
 *   SCCBF_OBS_CONE_BOX      the same record; CollisionCone2D.from_bounding_box (534-543):
 *                           a = hypot(extent x, extent y) + buffer, theta_o = 0, v_o = speed
 * Same answers as sccbf_ingest_bbox_* followed by the filter (ellipse rows bit for bit, cone rows to the
 * last few bits).
 */
#ifndef SCCBF_H_
#define SCCBF_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCCBF_VERSION 300 /* 0.3.0 */

/* obstacle kinds: 0..2 follow Obstacle2DTypes, reference cbf/obstacles.py:51-57 */
#define SCCBF_OBS_ELLIPSE 0
#define SCCBF_OBS_CONE 1
#define SCCBF_OBS_LANE 2
#define SCCBF_OBS_RADIAL 3 /* test_scripts/radial_dynamic_obstacles.py:391-411 */
#define SCCBF_OBS_ELLIPSE_PREP 4 /* Ellipse2D, reference cbf/obstacles.py:183-236, 310-317, constants evaluated */
#define SCCBF_OBS_CONE_PREP 5    /* CollisionCone2D, cbf/obstacles.py:401-502, obstacle velocity resolved */
#define SCCBF_OBS_ELLIPSE_BOX 6  /* Ellipse2D.from_bounding_box, cbf/obstacles.py:320-331, made from the box per tick */
#define SCCBF_OBS_CONE_BOX 7     /* CollisionCone2D.from_bounding_box, cbf/obstacles.py:534-543, likewise */
#define SCCBF_OBS_LANE6 8        /* PolyLane with more than four coefficients, cbf/obstacles.py:578-593 */
#define SCCBF_OBS_EMPTY 255

/* vehicle model */
#define SCCBF_DYN_DBM 0 /* DBM_CBF_2DS,  reference cbf/cbf.py:112-220, u = [a, beta]  */
#define SCCBF_DYN_KBM 1 /* KBM_VC_CBF2D, reference cbf/cbf.py:33-110,  u = [v, omega] */
#define SCCBF_DYN_DUM 2 /* DUM_CBF_2DS, reference cbf/cbf.py:222-298, u = [a, omega]: g_c of 237-239
                           (a drives v, omega drives yaw), drift [v cos, v sin, 0, 0]; as written the
                           drift is declared 5x1 with four entries (243-245) and the class does not run:
                           here with the 4x1 drift.  No steering transform (247-298 apply none).
                           lanes_per_qp 0, 4 (M <= 64), 8 (M <= 128), 16, 32 or 64 */
#define SCCBF_DYN_SADBM 3 /* SADBM_CBF_2DS, reference cbf/cbf.py:301-437: slip angle beta is a fifth
                             state, u = [a, d(beta)/dt].  Only through sccbf_solve_sa_* */

/* per-agent result */
#define SCCBF_STATUS_OPTIMAL 0
#define SCCBF_STATUS_INFEASIBLE 1 /* rows admit no u; u_out = u_ref (pass-through), or the least-violation
                                     input with SCCBF_FLAG_INFEASIBLE_MIN_VIOLATION */
#define SCCBF_STATUS_NONFINITE 2  /* NaN/Inf in the inputs; u_out = u_ref */

/* flags */
#define SCCBF_FLAG_OBS_SHARED 1u     /* obs is T[M][2][4], one table for all agents */
#define SCCBF_FLAG_TYPE_PER_AGENT 2u /* obs_type is u8[M][A] */
#define SCCBF_FLAG_SLIP_IO 4u        /* u_ref[1] / u_out[1] are already in QP coordinates (slip
                                        angle beta for DBM, yaw rate omega for KBM): skip the
                                        steering transforms of reference cbf/cbf.py:175,216,75,109 */
#define SCCBF_FLAG_OBS_TILED 8u      /* obs is T[ceil(A/64)][M][2][64][4]: tiles of 64 agents, each
                                        tile one contiguous block of 128 M 4-vectors; the last tile
                                        may be partly unused but must be allocated in full; not
                                        combinable with SCCBF_FLAG_OBS_SHARED */
#define SCCBF_FLAG_OBS_AGENT_MAJOR 128u /* obs is T[A][M][2][4]: the M slots of an agent side by side (slot m of agent a =
                                          the 8 scalars from obs + (a M + m) 8 on), the layout of a per-agent obstacle
                                          list.  The layout for long lists: the kernels for more than 128 rows
                                          give a workgroup a few agents, whose slots are then one contiguous
                                          stream (65536 agents, fp32: 5 % faster than from tiles at 257 rows, 8 % at
                                          512, 21 % at 1024; from tiles or planes eight agents share every 128-byte
                                          line).  Accepted wherever SCCBF_FLAG_OBS_TILED is; at most one of
                                          SCCBF_FLAG_OBS_SHARED, _TILED and _AGENT_MAJOR.  For lists of up to 64 rows
                                          tiles are the faster layout (12 % at 16 rows) */
#define SCCBF_FLAG_CONE_BETA_STATE 16u /* sccbf_solve_sa_* only: every cone slot takes beta from the
                                          agent's slip state instead of its own o[5], which is what
                                          the reference's cones hold after SADBM's first solve
                                          (cbf.py:424-427) */

#define SCCBF_FLAG_OBS_STREAMED 32u   /* the obstacle slots of this call will not be read again soon (every
                                          tick brings new ones): they are loaded with the non-temporal
                                          hint and leave the caches to the other arrays.  Without it the
                                          library decides by size.  A hint: results do not depend on it */

#define SCCBF_FLAG_INFEASIBLE_MIN_VIOLATION 64u /* sccbf_solve_* (with rows_out) and sccbf_qp_* only: an agent whose rows
                                          admit no input keeps status SCCBF_STATUS_INFEASIBLE, but u_out holds, instead of
                                          the nominal input, the input whose largest row violation is least, the
                                          violation of row i being (b_i - a_i . u) / sqrt(a_i R^-1 a_i^T), its distance
                                          to the half-plane in the metric of the cost; among such inputs the one
                                          nearest to the nominal input in that metric.  Rows with a zero normal and a
                                          positive right-hand side, which no input satisfies, are left out.  A
                                          second kernel after the filter, one thread per such agent, reading the
                                          rows from rows_out (sccbf_solve_*: must not be NULL) or rows (sccbf_qp_*).
                                          The reference's behaviour here is undefined: it applies whatever cvxopt
                                          returns for an infeasible program (cbf/cbf.py:213-220).  An agent for which
                                          the search does not end (48 certificates) keeps the nominal input */

/* error codes */
#define SCCBF_OK 0
#define SCCBF_ERR_ARG (-1)         /* NULL pointer / negative size / A >= 2^31 / bad enum or flags */
#define SCCBF_ERR_ROWS (-2)        /* M above SCCBF_MAX_ROWS */
#define SCCBF_ERR_WEIGHT (-3)      /* R not symmetric positive definite */
#define SCCBF_ERR_LAUNCH (-4)      /* hipLaunch failed */
#define SCCBF_ERR_VARIANT (-5)     /* lanes_per_qp cannot hold M rows */
#define SCCBF_ERR_NO_DEVICE (-6)   /* no HIP device / wrong architecture */
#define SCCBF_ERR_BATCHES (-7)     /* more batches than SCCBF_MAX_STATIC_BATCHES */

#define SCCBF_MAX_ROWS 1024

/*
 * Filter parameters; replaces the attributes of the reference filter objects:
 * alpha (cbf.py:130,147-148), lr/lf (150-152), L (61-62), R (154-157),
 * ZERO_TOL (cbf/utils.py:27).
 */
typedef struct sccbf_params {
    double alpha;         /* class-K gain, h_dot >= -alpha h */
    double lr, lf;        /* DBM: CoM to rear / front axle [m] */
    double L;             /* KBM: wheel base [m] */
    double R[4];          /* QP weight, row-major 2x2, symmetric positive definite */
    double zero_tol;      /* 1e-3 in the reference */
    int32_t dynamics;     /* SCCBF_DYN_* */
    int32_t lanes_per_qp; /* 0 = choose; 1,2,4,8,16 = register-resident rows with that
                             many cooperating lanes per QP (M <= 16 lanes, and <= 128); 16 with
                             M > 128, 32, 64 = rows staged in LDS only, any M <= SCCBF_MAX_ROWS whose
                             tables fit (64 / lanes tables of 3 M values per wavefront, 64 KB):
                             64 = one wavefront per QP.  Tables beyond 64 KB per workgroup need a
                             function attribute that the first such launch on a device sets
                             (hipFuncSetAttribute): make that launch outside stream capture, or
                             call sccbf_prewarm() first */
    uint32_t flags;       /* SCCBF_FLAG_* */
    int32_t reserved;
} sccbf_params;

int sccbf_version(void);
const char* sccbf_strerror(int code);

/*
 * Optional set-up of the current device, everything the library otherwise does on first use: the function attribute
 * that lets the kernels for long row lists (rows in LDS only, M > 128 or lanes_per_qp 32 / 64) take more than 64 KB
 * of LDS, for every such kernel, and the address of the table of sccbf_solve_batches_static.  Neither may happen
 * under stream capture: a caller that captures its first launch of a kind calls this once per device beforehand
 * (with that device current).  SCCBF_OK, or SCCBF_ERR_NO_DEVICE when the runtime refuses.  No counterpart in the
 * reference.  The library reads and clears no runtime error state: every entry point reports its own launch's
 * result (hipLaunchKernel's return value) and leaves hipGetLastError() to the caller.
 */
int sccbf_prewarm(void);

/* Writes the defaults of the reference objects: alpha 1 (cbf.py:128), R = I
 * (cbf.py:134), zero_tol 1e-3, lr = lf = 1.45, L = 2.9
 * (test_scripts/stanley_controller_ellipse.py:55-57), DBM, lanes_per_qp 0. */
void sccbf_default_params(sccbf_params* p);

/*
 * One control tick for A agents: barrier value and partials of every obstacle
 * slot, Lie derivatives, constraint rows, then the 2-input QP
 *      min (u - u_ref)^T R (u - u_ref)   s.t.   Lg_h . u >= -(Lf_h + alpha h + dh/dt)
 * including the steering <-> slip transforms on the way in and out.
 * Replaces DBM_CBF_2DS.solve_cbf (reference cbf/cbf.py:166-220) /
 * KBM_VC_CBF2D.solve_cbf (cbf.py:67-110) together with
 * ObstacleList2D.f/dx/dy/dtheta/dv/dt (cbf/obstacles.py:879-925) and the
 * per-obstacle formulas of Ellipse2D (183-236, 310-317), CollisionCone2D
 * (401-502), PolyLane (607-689) and the radial time-varying ellipse
 * (test_scripts/radial_dynamic_obstacles.py:391-411).
 */
int sccbf_solve_f32(const sccbf_params* p, int64_t A, int32_t M,
                    const float* state, const float* u_ref,
                    const float* obs, const uint8_t* obs_type,
                    float* u_out, int32_t* status, float* rows_out,
                    void* stream);
int sccbf_solve_f64(const sccbf_params* p, int64_t A, int32_t M,
                    const double* state, const double* u_ref,
                    const double* obs, const uint8_t* obs_type,
                    double* u_out, int32_t* status, double* rows_out,
                    void* stream);

/*
 * K independent batches of sccbf_solve_* in ONE launch: the same filter (DBM_CBF_2DS.solve_cbf, reference
 * cbf/cbf.py:166-220, KBM_VC_CBF2D.solve_cbf, 67-110, with the obstacle partials of cbf/obstacles.py named at
 * sccbf_solve_*), for callers whose agents arrive as separate batches, each with its own arrays: the
 * scenarios of a Monte-Carlo sweep, the clients of one serving tick.  A launch of 65536 agents x 16 rows ends
 * before its loads and its arithmetic have anything to overlap with (half of the memory rate); the same
 * agents as sixteen batches of one launch run at the rate of a 2^20-agent launch.  Answers are those of K
 * sccbf_solve_* calls, bit for bit.
 *   batches  DEVICE array of K records, owned by the caller and read by the kernel while it runs: per batch
 *            the number of agents and the arrays of sccbf_solve_* (rows_out optional).  Batches may differ
 *            in size.  The kernel SKIPS a record it cannot serve and writes none of its outputs: A <= 0,
 *            A > A_max, or a required pointer missing (a caller that fills status with a sentinel sees it).
 *   M, p     shared by all batches (p->flags say how every batch lays out obs and obs_type);
 *            dynamics SCCBF_DYN_DBM or SCCBF_DYN_KBM
 *   A_max    the largest A among the records (the host cannot read them): sizes the grid, < 2^31;
 *            0 = nothing to do
 *   K        0 ... SCCBF_MAX_BATCHES
 */
typedef struct sccbf_batch {
    int64_t A;
    const void* state;    /* T[A][4] */
    const void* u_ref;    /* T[A][2] */
    const void* obs;      /* layout by p->flags, M slots */
    const void* obs_type; /* u8[M] or u8[M][A] */
    void* u_out;          /* T[A][2] */
    void* status;         /* i32[A] */
    void* rows_out;       /* T[M][A][4] or NULL */
} sccbf_batch;

#define SCCBF_MAX_BATCHES 65535

int sccbf_solve_batches_f32(const sccbf_params* p, int32_t K, int32_t M, int64_t A_max,
                            const sccbf_batch* batches, void* stream);
int sccbf_solve_batches_f64(const sccbf_params* p, int32_t K, int32_t M, int64_t A_max,
                            const sccbf_batch* batches, void* stream);

/*
 * sccbf_solve_batches_* for callers that cannot allocate device memory for the table of records: code under
 * HIP graph capture, a tight serving loop, a harness that forbids hipMalloc.  No counterpart in the reference
 * (which has no device at all); same kernels and same answers as sccbf_solve_batches_*.
 *   batches       HOST array of K records (read during the call, the caller may reuse it at once); their
 *                 pointers are device pointers as everywhere.  Being host memory, the records are checked
 *                 here: A < 0, A >= 2^31 or a required pointer missing is SCCBF_ERR_ARG for the whole call.
 *   scalar_bytes  4: the arrays are float (as sccbf_solve_batches_f32), 8: double
 *   K             at most SCCBF_MAX_STATIC_BATCHES, else SCCBF_ERR_BATCHES
 * The records are staged in a table that the library itself holds in device memory: a file-scope array of
 * SCCBF_MAX_STATIC_BATCHES records (64 KB) on every GPU the library runs on, for the life of the process.  The
 * call allocates and frees nothing: it launches small kernels that write the records (passed as kernel
 * arguments) into the table, then the filter, all on `stream`.
 * All calls on one GPU share that table: calls on the same stream follow each other safely; calls whose launches
 * may overlap (different streams, or several host threads) race on it and must be serialised by the caller, or
 * use sccbf_solve_batches_* with tables of their own.  The first call on a device looks up the table's address
 * (hipGetSymbolAddress): make that call outside stream capture.
 */
#define SCCBF_MAX_STATIC_BATCHES 1024
int sccbf_solve_batches_static(const sccbf_params* p, int32_t scalar_bytes, int32_t K, int32_t M,
                               const sccbf_batch* batches, void* stream);

/*
 * Steer-rate model: SADBM_CBF_2DS.solve_cbf, reference cbf/cbf.py:349-437, for A agents.
 *   slip  T[A][2] in/out: beta (the fifth state, cbf.py:335,421) and the slip angle of the last
 *         nominal input, beta_ref_last (cbf.py:334,369); zeros before the first tick
 *   dt    the time step of this call [s] (cbf.py:332; the reference's dt=None mode measures wall
 *         clock, 361-367: such a caller measures on the host and passes the value, >= zero_tol)
 * Per agent: beta_ref = atan2(lr tan(delta_ref), lf + lr) (359), nominal input
 * [a_ref, (beta_ref - beta_ref_last) / dt] (367-372), rows
 *     [dh/dv, dh/dbeta] . u >= -(Lf_h + alpha h + dh/dt),
 *     Lf_h = dh/dx v cos(yaw + beta) + dh/dy v sin(yaw + beta) + dh/dyaw v sin(beta) / lr   (338-347, 398-405)
 * with dh/dbeta = dh/dyaw for a collision cone (obstacles.py:460-466) and 0 otherwise (124-125),
 * the QP, then beta += beta_dot* dt (421) and u_out = [a*, atan2((lf + lr) tan(beta), lr)] (430).
 * SCCBF_STATUS_INFEASIBLE: the nominal input is applied (beta += beta_dot_ref dt, u_out from it).
 * SCCBF_STATUS_NONFINITE: u_out = u_ref and the slip state is left as it was.
 * Cone slots read their own beta o[5] unless SCCBF_FLAG_CONE_BETA_STATE is set.
 * p->dynamics is ignored; SCCBF_FLAG_SLIP_IO is not allowed; M >= 1 (an empty list raises
 * in the reference, 375-378); rows_out as in sccbf_solve_*; lanes_per_qp 0, 4 (M <= 64), 8 (M <= 128), 16, 32 or 64.
 */
int sccbf_solve_sa_f32(const sccbf_params* p, double dt, int64_t A, int32_t M,
                       const float* state, float* slip, const float* u_ref,
                       const float* obs, const uint8_t* obs_type,
                       float* u_out, int32_t* status, float* rows_out, void* stream);
int sccbf_solve_sa_f64(const sccbf_params* p, double dt, int64_t A, int32_t M,
                       const double* state, double* slip, const double* u_ref,
                       const double* obs, const uint8_t* obs_type,
                       double* u_out, int32_t* status, double* rows_out, void* stream);

/*
 * The QP stage alone, on caller-supplied rows: for each agent
 *      min (u - u_nom)^T R (u - u_nom)   s.t.   rows[m][a][0..1] . u >= rows[m][a][2]
 * (rows is T[M][A][4], the 4th scalar is ignored; a row with a zero normal and
 * a non-positive right-hand side is "empty").  No steering transform is applied.
 * Replaces the cvxopt.solvers.cp call of reference cbf/cbf.py:213 (and 107).
 */
int sccbf_qp_f32(const sccbf_params* p, int64_t A, int32_t M,
                 const float* rows, const float* u_nom,
                 float* u_out, int32_t* status, void* stream);
int sccbf_qp_f64(const sccbf_params* p, int64_t A, int32_t M,
                 const double* rows, const double* u_nom,
                 double* u_out, int32_t* status, void* stream);

/*
 * Barrier value and partial derivatives of every obstacle slot, without rows or QP:
 * out is T[M][2][A][4] = (h, dh/dx, dh/dy, dh/dtheta | dh/dv, dh/dt, 0, 0).
 * Replaces the stacking primitives ObstacleList2D.f/dx/dy/dtheta/dv/dt (reference
 * cbf/obstacles.py:879-925) and the per-obstacle f/dx/dy/dtheta/dv/dt they call.
 */
int sccbf_partials_f32(const sccbf_params* p, int64_t A, int32_t M,
                       const float* state, const float* obs, const uint8_t* obs_type,
                       float* out, void* stream);
int sccbf_partials_f64(const sccbf_params* p, int64_t A, int32_t M,
                       const double* state, const double* obs, const uint8_t* obs_type,
                       double* out, void* stream);

/*
 * Nominal input in front of the filter, for A agents on one shared course.
 * Steering: LateralStanley.control, reference cbf/controllers.py:69-151 (the same law
 * is restated in test_scripts/stanley_controller_ellipse.py:146-212): nearest
 * way-point to the front axle over the WHOLE course (first index on ties), target
 * index never moving backwards, delta = normalize(yaw_des[idx] - yaw) +
 * atan2(k e, v + ks) with e the cross-track error at the nearest way-point.
 * Speed: PID1.control, cbf/controllers.py:173-180, on e = target_speed - v.
 */
typedef struct sccbf_nominal_params {
    double axle;         /* reference point -> front axle [m]: lf in LateralStanley
                            (controllers.py:80-81), the wheel base L in the driver scripts */
    double k, ks;        /* Stanley gains (controllers.py:144) */
    double kp, ki, kd;   /* PID1 gains */
    double dt;           /* PID1 time step (controllers.py:163,170-171) */
    double target_speed; /* [m/s] */
    uint32_t flags;      /* SCCBF_NOMINAL_* */
    int32_t reserved;
} sccbf_nominal_params;

#define SCCBF_NOMINAL_WAYPOINT_SPEED 1u /* track traj[idx][3] instead of target_speed */

/* Defaults of the reference objects: k 0.5, ks 0.01 (controllers.py:36-37), kp 1,
 * ki = kd = 0, dt 0.1 (155-163), axle 2.0 (lf, controllers.py:34), target 30 km/h
 * (stanley_controller_ellipse.py:590). */
void sccbf_default_nominal_params(sccbf_nominal_params* p);

/*
 *   state      T[A][4]   x, y, yaw, v
 *   traj       T[N][4]   way-points x, y, yaw_des, v_des (N >= 1), shared by all agents
 *   target_idx i32[A]    in: last target index (0 at the start); out: this tick's
 *   pid_state  T[A][2]   in/out: previous error, error integral; NULL = proportional
 *                        control only (a_ref = kp e)
 *   u_ref      T[A][2]   out: a_ref, delta_ref  (what sccbf_solve_* takes as u_ref)
 */
int sccbf_nominal_f32(const sccbf_nominal_params* p, int64_t A, int32_t N,
                      const float* state, const float* traj, int32_t* target_idx,
                      float* pid_state, float* u_ref, void* stream);
int sccbf_nominal_f64(const sccbf_nominal_params* p, int64_t A, int32_t N,
                      const double* state, const double* traj, int32_t* target_idx,
                      double* pid_state, double* u_ref, void* stream);

/*
 * Vehicle step of the closed-loop drivers: State.update_com, reference
 * test_scripts/stanley_controller_ellipse.py:122-131.  delta is clipped to
 * +-max_steer, beta = atan2(lr tan(delta), lf + lr),
 *   x += (v cos(yaw) - v sin(yaw) beta) dt,  y += (v sin(yaw) + v cos(yaw) beta) dt,
 *   yaw += (v beta / lr) dt,  v += a dt      (all from the values before the step).
 *   state T[A][4] in/out;  u T[A][2] = a, delta;  beta_out T[A] optional (may be NULL).
 */
int sccbf_vehicle_step_f32(int64_t A, float* state, const float* u, double lr, double lf,
                           double max_steer, double dt, float* beta_out, void* stream);
int sccbf_vehicle_step_f64(int64_t A, double* state, const double* u, double lr, double lf,
                           double max_steer, double dt, double* beta_out, void* stream);

/*
 * The kinematic steps of the reference drivers' State class, for the models whose input is not
 * (a, delta) of the dynamic bicycle; in place on state T[A][4], u T[A][2]:
 *   SCCBF_STEP_KIN_ACC   State.update, test_scripts/stanley_controller_ellipse.py:86-100: u = (a, delta),
 *                        x += v cos(yaw) dt, y += v sin(yaw) dt, yaw += v / L tan(delta) dt (normalised to
 *                        [-pi, pi], cbf/utils.py:93-106), v += a dt; delta clipped to +-max_steer
 *   SCCBF_STEP_KIN_VEL   State.update_by_vel, 102-118: u = (v, delta), the same with v = u[0] at the end
 *                        (what the drivers apply to KBM_VC_CBF2D's answer, 702-714)
 *   SCCBF_STEP_UNICYCLE  u = (a, omega): x, y as above, yaw += omega dt, v += a dt: the Euler step of the
 *                        model DUM_CBF_2DS filters (cbf/cbf.py:237-245); the reference ships no driver for it
 */
#define SCCBF_STEP_KIN_ACC 0
#define SCCBF_STEP_KIN_VEL 1
#define SCCBF_STEP_UNICYCLE 2
int sccbf_vehicle_step_kin_f32(int64_t A, float* state, const float* u, int32_t mode, double L,
                               double max_steer, double dt, void* stream);
int sccbf_vehicle_step_kin_f64(int64_t A, double* state, const double* u, int32_t mode, double L,
                               double max_steer, double dt, void* stream);

/*
 * One control tick's filter and vehicle step in one launch: sccbf_solve_* (reference
 * cbf/cbf.py:166-220) and then sccbf_vehicle_step_* (State.update_com, reference
 * test_scripts/stanley_controller_ellipse.py:122-131, called with the filtered input right after
 * the solve at 760-769) on the input just found.  Arguments as in those two: state T[A][4] is read
 * by the filter and then overwritten with the stepped state; u_out / status as in sccbf_solve_*;
 * beta_out T[A] optional.  Dynamics SCCBF_DYN_DBM without SCCBF_FLAG_SLIP_IO only (the step takes
 * a and delta); anything else: SCCBF_ERR_ARG.  Agents whose status is not optimal are stepped
 * with the nominal input that u_out then holds, as the two calls would.
 */
int sccbf_solve_step_f32(const sccbf_params* p, double max_steer, double dt, int64_t A, int32_t M,
                         float* state, const float* u_ref, const float* obs,
                         const uint8_t* obs_type, float* u_out, int32_t* status, float* beta_out,
                         void* stream);
int sccbf_solve_step_f64(const sccbf_params* p, double max_steer, double dt, int64_t A, int32_t M,
                         double* state, const double* u_ref, const double* obs,
                         const uint8_t* obs_type, double* u_out, int32_t* status, double* beta_out,
                         void* stream);

/*
 * Closed-loop rollout: `ticks` control ticks of
 *     nominal input (sccbf_nominal_*) -> filter (sccbf_solve_* / sccbf_solve_sa_*) -> vehicle step
 *     -> obstacle motion
 * for every agent in one launch; the main loop of the reference drivers
 * (test_scripts/stanley_controller_ellipse.py:630-760) for a whole batch.  Tick by tick
 * the results are those of the entry points called in turn.  By p->dynamics:
 *   SCCBF_DYN_DBM    u_ref = (a_ref, delta_ref); step sccbf_vehicle_step_* (update_com, 122-131), as the
 *                    drivers do after DBM_CBF_2DS.solve_cbf (760-769)
 *   SCCBF_DYN_KBM    u_ref = (v, delta_ref) with v the tracked speed (target_speed or the way-point's,
 *                    633); step SCCBF_STEP_KIN_VEL (update_by_vel, 102-118), as after
 *                    KBM_VC_CBF2D.solve_cbf (702-714); pid_state is carried but does not enter u_ref
 *   SCCBF_DYN_DUM    u_ref = (a_ref, v tan(delta_ref) / L), the yaw rate of the nominal steering (the
 *                    relation of cbf/cbf.py:75); step SCCBF_STEP_UNICYCLE.  The reference ships no
 *                    driver for this model: the loop is this library's
 *   SCCBF_DYN_SADBM  sccbf_solve_sa_* with dt = the tick length and SCCBF_FLAG_CONE_BETA_STATE (every
 *                    cone takes the agent's slip angle, cbf/cbf.py:424-427), slip carried in `slip`;
 *                    step update_com with the filter's (a, delta), as the drivers' commented-out
 *                    SADBM lines would (734, 807)
 */
#define SCCBF_MOTION_STATIC 0            /* obstacle slots stay where they are */
#define SCCBF_MOTION_CONSTANT_VELOCITY 1 /* per tick: ellipse centre += (vx, vy) dt; cone centre
                                            += v_o (cos, sin)(theta_o + beta) dt; radial centre
                                            += (cdot_x, cdot_y) dt; lanes stay */
#define SCCBF_MOTION_SEEKER 2            /* ObstacleSpawner.update_seekers, reference
                                            test_scripts/radial_dynamic_obstacles.py:193-239: after
                                            the vehicle step every ellipse and radial slot turns
                                            toward the agent's new position, yaw = atan2(y - cy,
                                            x - cx), takes the speed max(seek_gain * distance,
                                            seek_v_min) along that yaw (ellipse: theta_o = yaw too,
                                            obstacles.py:288-292) and moves by velocity * dt; cones
                                            keep their constant velocity, lanes stay */
typedef struct sccbf_rollout_params {
    double max_steer; /* steering clip of the vehicle step [rad] */
    double dt;        /* tick length [s] */
    int32_t ticks;
    int32_t motion;   /* SCCBF_MOTION_* */
    double seek_gain;  /* SCCBF_MOTION_SEEKER: speed per metre of distance [1/s] */
    double seek_v_min; /* SCCBF_MOTION_SEEKER: least speed [m/s] */
} sccbf_rollout_params;

/* max_steer 30 deg, dt 0.1 (stanley_controller_ellipse.py:54,58), 0 ticks, static obstacles,
 * seek_gain 0.2, seek_v_min 3.0 (radial_dynamic_obstacles.py:193) */
void sccbf_default_rollout_params(sccbf_rollout_params* p);

/*
 *   state        T[A][4]   in: initial states; out: states after the last tick
 *   traj         T[N][4]   course, N >= 1 way-points: up to 2048 stay in LDS for the whole launch, a longer
 *                          course is staged pass by pass on every tick (slower).  N = 0
 *                          (traj, target_idx, pid_state then unused, may be NULL): no course, the
 *                          nominal input is (0, 0) on every tick, as for the stationary ego of
 *                          radial_dynamic_obstacles.py:444
 *   target_idx   i32[A]    in/out, as in sccbf_nominal_*
 *   pid_state    T[A][2]   in/out or NULL, as in sccbf_nominal_*
 *   slip         T[A][2]   in/out, SCCBF_DYN_SADBM only (else unused, may be NULL): beta, beta_ref_last
 *                          as in sccbf_solve_sa_*
 *   obs, obs_type          as in sccbf_solve_* (planes or tiles; not the shared table), M >= 1: up to 64
 *                          slots four lanes serve an agent, beyond that a whole wavefront does;
 *                          read once: moving obstacles advance in registers, obs is not written
 *   h_min        T[A]      optional: least barrier value over all ticks and slots
 *   n_active     i32[A]    optional: ticks on which the filter changed the nominal input
 *   n_infeasible i32[A]    optional: ticks with SCCBF_STATUS_INFEASIBLE
 *   trace        T[ticks][A][4] optional: per tick the filter's answer u[0], u[1] (in the model's input
 *                          coordinates, as u_out of the solve entry), the slip angle beta of an update_com
 *                          step (0 for the other steps), status
 * p->flags: the layout flags only (SCCBF_FLAG_SLIP_IO, SCCBF_FLAG_CONE_BETA_STATE: SCCBF_ERR_ARG).
 */
int sccbf_rollout_f32(const sccbf_params* p, const sccbf_nominal_params* np,
                      const sccbf_rollout_params* rp, int64_t A, int32_t M, int32_t N,
                      float* state, const float* traj, int32_t* target_idx, float* pid_state,
                      float* slip, const float* obs, const uint8_t* obs_type,
                      float* h_min, int32_t* n_active, int32_t* n_infeasible, float* trace,
                      void* stream);
int sccbf_rollout_f64(const sccbf_params* p, const sccbf_nominal_params* np,
                      const sccbf_rollout_params* rp, int64_t A, int32_t M, int32_t N,
                      double* state, const double* traj, int32_t* target_idx, double* pid_state,
                      double* slip, const double* obs, const uint8_t* obs_type,
                      double* h_min, int32_t* n_active, int32_t* n_infeasible, double* trace,
                      void* stream);

/*
 * Bounding boxes -> obstacle slots, for A agents with M boxes each: what
 * ObstacleList2D.update_by_bounding_box (reference cbf/obstacles.py:833-858) builds per box.
 *   bbox  T[M][2][A][4]  location x, y, extent x, y | yaw, speed, -, -   (BoundingBox, 59-64)
 *   kind  u8[M] (u8[M][A] with SCCBF_FLAG_TYPE_PER_AGENT), afterwards the obs_type of the filter:
 *         SCCBF_OBS_ELLIPSE  Ellipse2D.from_bounding_box (320-331): a = extent x + buffer,
 *                            b = extent y + buffer, theta_o = yaw, at rest
 *         SCCBF_OBS_CONE     CollisionCone2D.from_bounding_box (534-543): a = hypot(extent x,
 *                            extent y) + buffer, theta_o = 0, v_o = speed
 *         SCCBF_OBS_ELLIPSE_PREP, SCCBF_OBS_CONE_PREP   the same two in their prepared forms
 *         anything else      an all-zero slot (pair it with SCCBF_OBS_EMPTY)
 *   obs   out, T[M][2][A][4], or tiles / per-agent lists with SCCBF_FLAG_OBS_TILED / _AGENT_MAJOR in flags
 */
int sccbf_ingest_bbox_f32(int64_t A, int32_t M, const float* bbox, const uint8_t* kind,
                          double buffer, uint32_t flags, float* obs, void* stream);
int sccbf_ingest_bbox_f64(int64_t A, int32_t M, const double* bbox, const uint8_t* kind,
                          double buffer, uint32_t flags, double* obs, void* stream);

/*
 * As-written ellipse and cone slots -> their prepared forms (SCCBF_OBS_ELLIPSE_PREP,
 * SCCBF_OBS_CONE_PREP), in place: the values that reference cbf/obstacles.py recomputes from the
 * obstacle alone in every partial of every tick (np.cos / np.sin of the ellipse angle, 1/a^2,
 * 1/b^2: 183-236; the cone's obstacle velocity v_o (cos, sin)(theta_o + beta): 489-490) are
 * evaluated once, in double precision, and obs_type is rewritten to the prepared kinds.  Other
 * kinds and already prepared slots are left as they are.  Call it when obstacles change, not per
 * tick; sccbf_ingest_bbox_* writes prepared slots directly when `kind` names them.
 *   obs       in/out, layout by flags: planes, SCCBF_FLAG_OBS_TILED, _AGENT_MAJOR or _SHARED
 *   obs_type  in/out, u8[M] or u8[M][A] with SCCBF_FLAG_TYPE_PER_AGENT
 * A prepared cone keeps the beta it was prepared with (SCCBF_FLAG_CONE_BETA_STATE does not
 * reach it).
 */
int sccbf_prepare_obs_f32(int64_t A, int32_t M, float* obs, uint8_t* obs_type, uint32_t flags,
                          void* stream);
int sccbf_prepare_obs_f64(int64_t A, int32_t M, double* obs, uint8_t* obs_type, uint32_t flags,
                          void* stream);

/*
 * Lane points -> lane slot m of every agent: PolyLane.fit_polynomial_curve, reference
 * cbf/obstacles.py:715-773, the polynomial of degree 0..3 that minimises
 * sum(((p(x_k) - y_k) / sigma_k)^2).
 *   px, py  T[K][A]   K >= 1 points per agent
 *   sigma   T[K][A]   standard deviations, or NULL for 10 everywhere (737-738); the reference's
 *                     fixed points are points with a small sigma (alpha = 0.01, 753-756); a
 *                     point with sigma <= 0 or a non-finite value is skipped (ragged counts)
 *   obs     in/out: only slot m (0 <= m < M) is written, c0..c3 | buffer, 0, 0, 0; layout as in
 *           sccbf_solve_* (planes, or SCCBF_FLAG_OBS_TILED / _AGENT_MAJOR in flags)
 * Fewer usable points than coefficients, or points that do not determine them: NaN coefficients
 * (the filter then reports SCCBF_STATUS_NONFINITE for that agent).
 */
int sccbf_lane_fit_f32(int64_t A, int32_t K, int32_t degree, const float* px, const float* py,
                       const float* sigma, double buffer, int32_t m, int32_t M, uint32_t flags,
                       float* obs, void* stream);
int sccbf_lane_fit_f64(int64_t A, int32_t K, int32_t degree, const double* px, const double* py,
                       const double* sigma, double buffer, int32_t m, int32_t M, uint32_t flags,
                       double* obs, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SCCBF_H_ */
