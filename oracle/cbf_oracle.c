/*
 * cbf_oracle.c - CPU restatement (plain C, fp64) of the reference's CBF-QP
 * safety-filter arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker and the timed CPU
 * baseline; it is never on the product path.  Only tests/, the smoke check
 * and the cpu_baseline leg of bench.py may load it.
 *
 * Parity pin: the closed-loop replay of the reference's as-shipped scenario
 * (tests/test_oracle_golden.py) reproduces all 277 samples of the reference's
 * saved test_scripts/beta_vs_time.mat (tests/golden/beta_vs_time.npz), which
 * contains the answers of the reference's own cvxopt solve, for the
 * single-row collision-cone case.  Multi-row QPs and the ellipse / radial /
 * lane rows have no reference-held numbers: for those this oracle is checked
 * against a second, independent algorithm in this file, against SciPy's SLSQP
 * and by KKT residuals ("parity unpinned" against cvxopt itself, which is not
 * installed here).  The rows of the unicycle-input and steer-rate models
 * (DUM_CBF_2DS, SADBM_CBF_2DS) are restated from the reference's formulas and
 * checked against the matrix products it writes down; the reference holds no
 * numbers for them (DUM does not run as shipped, SADBM is commented out in its
 * drivers): parity unpinned.
 *
 * Every function cites the reference lines (relative to /root/reference) whose
 * arithmetic it restates.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>

#define ORC_OBS_ELLIPSE 0
#define ORC_OBS_CONE 1
#define ORC_OBS_LANE 2
#define ORC_OBS_RADIAL 3
#define ORC_OBS_ELLIPSE_PREP 4
#define ORC_OBS_CONE_PREP 5
#define ORC_OBS_ELLIPSE_BOX 6
#define ORC_OBS_CONE_BOX 7
#define ORC_OBS_LANE6 8
#define ORC_OBS_EMPTY 255

#define ORC_DYN_DBM 0
#define ORC_DYN_KBM 1
#define ORC_DYN_DUM 2
#define ORC_DYN_SADBM 3

#define ORC_FLAG_OBS_SHARED 1u
#define ORC_FLAG_TYPE_PER_AGENT 2u
#define ORC_FLAG_SLIP_IO 4u
#define ORC_FLAG_CONE_BETA_STATE 16u

#define ORC_LANE_ITERS 32
#define ORC_LANE_HALVINGS 30

/* same field order as sccbf_params in include/sccbf.h */
typedef struct orc_params {
    double alpha;
    double lr, lf;
    double L;
    double R[4];
    double zero_tol;
    int32_t dynamics;
    int32_t lanes_per_qp; /* unused here */
    uint32_t flags;
    int32_t reserved;
} orc_params;

/* h and its partials w.r.t. ego x, y, yaw, speed and time */
typedef struct orc_partials {
    double h, hx, hy, hth, hv, ht;
} orc_partials;

/* ------------------------------------------------------------------ barriers */

/* Ellipse2D.evaluate/dx/dy/dv/dtheta/dt, cbf/obstacles.py:183-194, 210-236,
 * 304-317.  o = cx, cy, a, b, theta_o, vx, vy.  dt() is reproduced as written:
 * it ignores the rotation of the ellipse. */
static void orc_ellipse(const double s[4], const double o[8], orc_partials* r)
{
    double xd = s[0] - o[0], yd = s[1] - o[1];
    double a = o[2], b = o[3];
    double ct = cos(o[4]), st = sin(o[4]);
    double p = xd * ct + yd * st;
    double q = -xd * st + yd * ct;
    r->h = (p / a) * (p / a) + (q / b) * (q / b) - 1.0;
    r->hx = (2.0 * ct / (a * a)) * p + (-2.0 * st / (b * b)) * q;
    r->hy = (2.0 * st / (a * a)) * p + (2.0 * ct / (b * b)) * q;
    r->hth = 0.0;
    r->hv = 0.0;
    r->ht = -2.0 * ((xd / (a * a)) * o[5] + (yd / (b * b)) * o[6]);
}

/* The same Ellipse2D formulas (cbf/obstacles.py:183-194, 210-236, 304-317) on a slot that
 * already holds the per-obstacle constants: o = cx, cy, 1/a^2, 1/b^2, cos(theta_o),
 * sin(theta_o), vx, vy. */
static void orc_ellipse_prep(const double s[4], const double o[8], orc_partials* r)
{
    double xd = s[0] - o[0], yd = s[1] - o[1];
    double ia2 = o[2], ib2 = o[3], ct = o[4], st = o[5];
    double p = xd * ct + yd * st;
    double q = -xd * st + yd * ct;
    r->h = p * p * ia2 + q * q * ib2 - 1.0;
    r->hx = (2.0 * ct * ia2) * p + (-2.0 * st * ib2) * q;
    r->hy = (2.0 * st * ia2) * p + (2.0 * ct * ib2) * q;
    r->hth = 0.0;
    r->hv = 0.0;
    r->ht = -2.0 * ((xd * ia2) * o[6] + (yd * ib2) * o[7]);
}

/* Time-varying ellipse with a speed-dependent margin,
 * test_scripts/radial_dynamic_obstacles.py:391-405.
 * o = cx, cy, a, b, cdot_x, cdot_y, kv. */
static void orc_radial(const double s[4], const double o[8], orc_partials* r)
{
    double xd = s[0] - o[0], yd = s[1] - o[1];
    double a = o[2], b = o[3], kv = o[6], v = s[3];
    r->h = (xd / a) * (xd / a) + (yd / b) * (yd / b) - 1.0 - (kv * v / (1.0 + v));
    r->hx = 2.0 * xd / (a * a);
    r->hy = 2.0 * yd / (b * b);
    r->hth = 0.0;
    r->hv = -kv / ((1.0 + v) * (1.0 + v));
    r->ht = -2.0 * ((xd / (a * a)) * o[4] + (yd / (b * b)) * o[5]);
}

/* CollisionCone2D.update/evaluate/dx/dy/dv/dtheta/dt,
 * cbf/obstacles.py:468-502, 401-458.  o = cx, cy, theta_o, v_o, a, beta. */
static void orc_cone_core(const double s[4], double cx, double cy, double ovx, double ovy, double a,
                          double beta, double eps, orc_partials* r)
{
    double xd = s[0] - cx, yd = s[1] - cy;
    double svx = s[3] * cos(s[2]), svy = s[3] * sin(s[2]);
    double dvx = svx - ovx, dvy = svy - ovy;
    double dist = sqrt(xd * xd + yd * yd);
    double vn = sqrt(dvx * dvx + dvy * dvy);
    double cb = eps;
    if (fabs(dist) > fabs(a)) cb = sqrt(dist * dist - a * a) + eps;
    double cos_phi = (dist > eps) ? cb / dist : 0.0;
    double c = cos(s[2] + beta), sn = sin(s[2] + beta);
    r->h = (xd * dvx + yd * dvy) + dist * vn * cos_phi;
    r->hx = dvx + vn * xd / (cb + eps);
    r->hy = dvy + vn * yd / (cb + eps);
    r->hv = (xd * c + yd * sn) + (dvx * c + dvy * sn) * cb / (vn + eps);
    r->hth = (-xd * svy + yd * svx) + (-dvx * svy + dvy * svx) * cb / (vn + eps);
    r->ht = (-dvx * ovx - dvy * ovy) - vn * (xd * ovx + yd * ovy) / (cb + eps);
}

static void orc_cone(const double s[4], const double o[8], double eps, orc_partials* r)
{
    double beta = o[5];
    orc_cone_core(s, o[0], o[1], o[3] * cos(o[2] + beta), o[3] * sin(o[2] + beta), o[4], beta, eps, r);
}

/* The same collision cone on a slot that holds the obstacle's velocity already resolved
 * (cbf/obstacles.py:489-490): o = cx, cy, v_o cos(theta_o + beta), v_o sin(theta_o + beta), a, beta. */
static void orc_cone_prep(const double s[4], const double o[8], double eps, orc_partials* r)
{
    orc_cone_core(s, o[0], o[1], o[2], o[3], o[4], o[5], eps, r);
}

static void orc_poly3(const double c[4], double x, double* f, double* df, double* ddf)
{
    *f = c[0] + x * (c[1] + x * (c[2] + x * c[3]));
    *df = c[1] + x * (2.0 * c[2] + x * 3.0 * c[3]);
    *ddf = 2.0 * c[2] + 6.0 * c[3] * x;
}

/* PolyLane.get_shortest_distance_x, cbf/obstacles.py:641-679: abscissa of the
 * curve point nearest to (px, py), found by descent from x0 = px.  The
 * reference hands the problem to scipy's Newton-CG (with a mis-stated Hessian,
 * 669-673, which changes the path but not a converged stationary point).  Here:
 * Newton with the true second derivative, steepest-descent scaling where the
 * curvature is not positive, and step halving until the distance does not
 * increase; a fixed iteration budget.  For lines (c2 = c3 = 0) the first step
 * is exact. */
double orc_lane_closest_x(const double c[4], double px, double py)
{
    double x = px;
    for (int it = 0; it < ORC_LANE_ITERS; ++it) {
        double f, df, ddf;
        orc_poly3(c, x, &f, &df, &ddf);
        double rr = f - py;
        double g0 = (x - px) * (x - px) + rr * rr;
        double g1 = 2.0 * (x - px) + 2.0 * rr * df;
        double g2 = 2.0 * (1.0 + df * df + rr * ddf);
        double step = (g2 > 1e-12) ? -g1 / g2 : -g1 / (fabs(g2) + 1.0);
        if (step == 0.0) break;
        double xn = x + step;
        double glim = g0 * (1.0 + 8.0 * 2.220446049250313e-16);
        for (int k = 0; k < ORC_LANE_HALVINGS; ++k) {
            double fn, dfn, ddfn;
            orc_poly3(c, xn, &fn, &dfn, &ddfn);
            double gn = (xn - px) * (xn - px) + (fn - py) * (fn - py);
            if (gn <= glim) break;
            step *= 0.5;
            xn = x + step;
        }
        x = xn;
        /* quadratic convergence: a step this small leaves an error below rounding */
        if (fabs(step) <= 1e-8 * (1.0 + fabs(x))) break;
    }
    return x;
}

/* The same for a polynomial of any degree below ORC_LANE_MAX_COEFFS coefficients: PolyLane takes a coefficient
 * array of any length (cbf/obstacles.py:578-593, numpy Polynomial and its two derivatives). */
#define ORC_LANE_MAX_COEFFS 7
static void orc_polyn(const double* c, int n, double x, double* f, double* df, double* ddf)
{
    double a = 0.0, b = 0.0, d = 0.0;
    for (int k = n - 1; k >= 0; --k) a = c[k] + x * a;
    for (int k = n - 1; k >= 1; --k) b = k * c[k] + x * b;
    for (int k = n - 1; k >= 2; --k) d = (double)(k * (k - 1)) * c[k] + x * d;
    *f = a;
    *df = b;
    *ddf = d;
}

double orc_lane_closest_x_n(const double* c, int n, double px, double py)
{
    double x = px;
    for (int it = 0; it < ORC_LANE_ITERS; ++it) {
        double f, df, ddf;
        orc_polyn(c, n, x, &f, &df, &ddf);
        double rr = f - py;
        double g0 = (x - px) * (x - px) + rr * rr;
        double g1 = 2.0 * (x - px) + 2.0 * rr * df;
        double g2 = 2.0 * (1.0 + df * df + rr * ddf);
        double step = (g2 > 1e-12) ? -g1 / g2 : -g1 / (fabs(g2) + 1.0);
        if (step == 0.0) break;
        double xn = x + step;
        double glim = g0 * (1.0 + 8.0 * 2.220446049250313e-16);
        for (int k = 0; k < ORC_LANE_HALVINGS; ++k) {
            double fn, dfn, ddfn;
            orc_polyn(c, n, xn, &fn, &dfn, &ddfn);
            double gn = (xn - px) * (xn - px) + (fn - py) * (fn - py);
            /* a step already below the tolerance is taken as it is: the two distances differ by less than
             * their own rounding, and halving the step on that noise would leave x one step short */
            if (gn <= glim || fabs(step) <= 1e-8 * (1.0 + fabs(x))) break;
            step *= 0.5;
            xn = x + step;
        }
        x = xn;
        if (fabs(step) <= 1e-8 * (1.0 + fabs(x))) break;
    }
    return x;
}

/* A lane of degree up to 6: o = c0 ... c6, buffer.  Same formulas as orc_lane below. */
static void orc_lane6(const double s[4], const double o[8], double eps, orc_partials* r)
{
    double cx = orc_lane_closest_x_n(o, ORC_LANE_MAX_COEFFS, s[0], s[1]);
    double g, dg, ddg;
    orc_polyn(o, ORC_LANE_MAX_COEFFS, cx, &g, &dg, &ddg);
    double eta = 1.0 + dg * ddg + dg * dg - s[1] * ddg;
    if (fabs(eta) < eps) eta = eps;
    r->h = (cx - s[0]) * (cx - s[0]) + (g - s[1]) * (g - s[1]) - o[7];
    r->hx = (2.0 / eta) * ((s[0] - cx) * (eta - 1.0) - (s[1] - g) * dg);
    r->hy = (2.0 / eta) * (-(s[0] - cx) * dg + (s[1] - g) * (eta - dg * dg));
    r->hth = 0.0;
    r->hv = 0.0;
    r->ht = 0.0;
}

/* PolyLane.update/evaluate/dx/dy, cbf/obstacles.py:620-636, 607-612, 681-689.
 * o = c0, c1, c2, c3, buffer.  eta and the un-squared buffer are as written. */
static void orc_lane(const double s[4], const double o[8], double eps, orc_partials* r)
{
    double cx = orc_lane_closest_x(o, s[0], s[1]);
    double g, dg, ddg;
    orc_poly3(o, cx, &g, &dg, &ddg);
    double eta = 1.0 + dg * ddg + dg * dg - s[1] * ddg;
    if (fabs(eta) < eps) eta = eps;
    r->h = (cx - s[0]) * (cx - s[0]) + (g - s[1]) * (g - s[1]) - o[4];
    r->hx = (2.0 / eta) * ((s[0] - cx) * (eta - 1.0) - (s[1] - g) * dg);
    r->hy = (2.0 / eta) * (-(s[0] - cx) * dg + (s[1] - g) * (eta - dg * dg));
    r->hth = 0.0;
    r->hv = 0.0;
    r->ht = 0.0;
}

/* Box kinds: the slot holds a bounding box (reference cbf/obstacles.py:59-64: location x, y, extent x, y, yaw,
 * speed) and the buffer; Ellipse2D.from_bounding_box (320-331: a = extent x + buffer, b = extent y + buffer,
 * theta = yaw, at rest) or CollisionCone2D.from_bounding_box (534-543: a = hypot(extent) + buffer,
 * s_obs = [x, y, 0, speed]) makes the obstacle.  In place: type and slot become the as-written kind. */
static void orc_unbox(int* type, double o[8])
{
    if (*type == ORC_OBS_ELLIPSE_BOX) {
        const double q[8] = {o[0], o[1], o[2] + o[6], o[3] + o[6], o[4], 0.0, 0.0, 0.0};
        memcpy(o, q, sizeof(q));
        *type = ORC_OBS_ELLIPSE;
    } else if (*type == ORC_OBS_CONE_BOX) {
        const double q[8] = {o[0], o[1], 0.0, o[5], hypot(o[2], o[3]) + o[6], 0.0, 0.0, 0.0};
        memcpy(o, q, sizeof(q));
        *type = ORC_OBS_CONE;
    }
}

/* returns 0 for an empty slot (no row), 1 otherwise */
int orc_barrier(int type, const double s[4], const double o_in[8], double zero_tol, orc_partials* r)
{
    double o[8];
    memcpy(o, o_in, sizeof(o));
    orc_unbox(&type, o);
    switch (type) {
    case ORC_OBS_ELLIPSE: orc_ellipse(s, o, r); return 1;
    case ORC_OBS_CONE: orc_cone(s, o, zero_tol, r); return 1;
    case ORC_OBS_LANE: orc_lane(s, o, zero_tol, r); return 1;
    case ORC_OBS_LANE6: orc_lane6(s, o, zero_tol, r); return 1;
    case ORC_OBS_RADIAL: orc_radial(s, o, r); return 1;
    case ORC_OBS_ELLIPSE_PREP: orc_ellipse_prep(s, o, r); return 1;
    case ORC_OBS_CONE_PREP: orc_cone_prep(s, o, zero_tol, r); return 1;
    default:
        memset(r, 0, sizeof(*r));
        r->h = 1.0;
        return 0;
    }
}

/* ------------------------------------------------------------ rows (Lie derivatives) */

/* Row Lg_h . u >= b from the partials.
 * DBM: g_c, f_c of cbf/cbf.py:159-164 and the contraction 200-207:
 *   Lg_h = [ h_v ,  v(-sin(th) h_x + cos(th) h_y) + (v/lr) h_th ]
 *   Lf_h = v (cos(th) h_x + sin(th) h_y),   b = -(Lf_h + alpha h + h_t)
 * KBM: g_c of cbf/cbf.py:94, rows 98-101 (no drift, no h_t):
 *   Lg_h = [ cos(th) h_x + sin(th) h_y ,  h_th ],   b = -alpha h
 * DUM: g_c of cbf/cbf.py:237-239 (column lists: a -> v, omega -> yaw), drift 243-245 with its
 * four entries (the declared 5x1 shape does not build), contraction 279-286:
 *   Lg_h = [ h_v , h_th ],   Lf_h = v (cos(th) h_x + sin(th) h_y),   b as for DBM */
void orc_row(const orc_params* p, const double s[4], const orc_partials* d, double row[3])
{
    double ct = cos(s[2]), st = sin(s[2]), v = s[3];
    if (p->dynamics == ORC_DYN_KBM) {
        row[0] = ct * d->hx + st * d->hy;
        row[1] = d->hth;
        row[2] = -(p->alpha * d->h);
    } else if (p->dynamics == ORC_DYN_DUM) {
        row[0] = d->hv;
        row[1] = d->hth;
        double lf_h = v * (ct * d->hx + st * d->hy);
        row[2] = -(lf_h + p->alpha * d->h + d->ht);
    } else {
        row[0] = d->hv;
        row[1] = v * (-st * d->hx + ct * d->hy) + (v / p->lr) * d->hth;
        double lf_h = v * (ct * d->hx + st * d->hy);
        row[2] = -(lf_h + p->alpha * d->h + d->ht);
    }
}

/* SADBM_CBF_2DS, cbf/cbf.py:338-347 (g_c, f_c) and 398-405: state x, y, yaw, v, beta;
 *   Lg_h = [ h_v , h_beta ],
 *   Lf_h = h_x v cos(yaw + beta) + h_y v sin(yaw + beta) + h_th v sin(beta) / lr
 * h_beta: CollisionCone2D.dbeta = dtheta (cbf/obstacles.py:460-466), 0 for the other kinds
 * (Obstacle2DBase.dbeta, obstacles.py:124-125). */
void orc_row_sa(const orc_params* p, const double s[4], double beta, const orc_partials* d,
                double h_beta, double row[3])
{
    double v = s[3];
    double lf_h = d->hx * v * cos(s[2] + beta) + d->hy * v * sin(s[2] + beta) +
                  d->hth * v * sin(beta) / p->lr;
    row[0] = d->hv;
    row[1] = h_beta;
    row[2] = -(lf_h + p->alpha * d->h + d->ht);
}

/* ------------------------------------------------------------------- the QP */

/* Cholesky factor of the symmetrised weight: R = Lc Lc^T, Lc = [[l00,0],[l10,l11]].
 * returns 0 when R is not positive definite. */
static int orc_chol2(const double R[4], double* l00, double* l10, double* l11)
{
    double r00 = R[0], r01 = 0.5 * (R[1] + R[2]), r11 = R[3];
    if (!(r00 > 0.0)) return 0;
    *l00 = sqrt(r00);
    *l10 = r01 / *l00;
    double t = r11 - (*l10) * (*l10);
    if (!(t > 0.0)) return 0;
    *l11 = sqrt(t);
    return 1;
}

/* In w = Lc^T (u - u_nom) the problem is: least-norm point of the polygon
 * { at_i . w >= bt_i }.  Rows are scaled to unit normals; a row with a zero
 * normal is flagged with at = (0,0) and keeps its right-hand side. */
static void orc_to_wspace(int M, const double* rows /*[M][3]*/, const double u_nom[2],
                          double l00, double l10, double l11, double* at0, double* at1, double* bt)
{
    for (int i = 0; i < M; ++i) {
        double a0 = rows[3 * i], a1 = rows[3 * i + 1], b = rows[3 * i + 2];
        double t0 = a0 / l00;
        double t1 = (a1 - l10 * t0) / l11;
        double rhs = b - (a0 * u_nom[0] + a1 * u_nom[1]);
        double n = sqrt(t0 * t0 + t1 * t1);
        if (n > 1e-150) {
            at0[i] = t0 / n;
            at1[i] = t1 / n;
            bt[i] = rhs / n;
        } else {
            at0[i] = 0.0;
            at1[i] = 0.0;
            bt[i] = rhs;
        }
    }
}

static void orc_from_wspace(const double w[2], const double u_nom[2], double l00, double l10,
                            double l11, double u[2])
{
    /* u = u_nom + Lc^-T w */
    double y1 = w[1] / l11;
    double y0 = (w[0] - l10 * y1) / l00;
    u[0] = u_nom[0] + y0;
    u[1] = u_nom[1] + y1;
}

static int orc_feasible(int M, const double* at0, const double* at1, const double* bt,
                        double w0, double w1, double tol)
{
    for (int j = 0; j < M; ++j) {
        double lhs = at0[j] * w0 + at1[j] * w1;
        double scale = fabs(at0[j] * w0) + fabs(at1[j] * w1) + fabs(bt[j]) + 1e-300;
        if (lhs - bt[j] < -tol * scale) return 0;
    }
    return 1;
}

/* Exact solve by active-set enumeration: the minimiser of a strictly convex
 * quadratic over a polygon is the unconstrained minimiser, its projection on
 * one edge line, or a vertex; the feasible candidate of least cost is the
 * answer.  Stands in for cvxopt.solvers.cp of cbf/cbf.py:213 (and 107).
 * rows: [M][3] = a0, a1, b with a . u >= b.  returns status 0 / 1. */
int orc_qp2_enum(int M, const double* rows, const double R[4], const double u_nom[2],
                 double u[2], double* work /* 3*M doubles */)
{
    double l00, l10, l11;
    u[0] = u_nom[0];
    u[1] = u_nom[1];
    if (!orc_chol2(R, &l00, &l10, &l11)) return 1;
    double *at0 = work, *at1 = work + M, *bt = work + 2 * M;
    orc_to_wspace(M, rows, u_nom, l00, l10, l11, at0, at1, bt);
    const double tol = 1e-10;
    for (int i = 0; i < M; ++i)
        if (at0[i] == 0.0 && at1[i] == 0.0 && bt[i] > 0.0) return 1;

    double best = INFINITY, bw0 = 0.0, bw1 = 0.0;
    if (orc_feasible(M, at0, at1, bt, 0.0, 0.0, tol)) {
        best = 0.0;
    } else {
        for (int i = 0; i < M; ++i) {
            if (at0[i] == 0.0 && at1[i] == 0.0) continue;
            double w0 = at0[i] * bt[i], w1 = at1[i] * bt[i];
            double c = w0 * w0 + w1 * w1;
            if (c < best && orc_feasible(M, at0, at1, bt, w0, w1, tol)) {
                best = c; bw0 = w0; bw1 = w1;
            }
        }
        for (int i = 0; i < M; ++i) {
            for (int j = i + 1; j < M; ++j) {
                double det = at0[i] * at1[j] - at1[i] * at0[j];
                if (fabs(det) < 1e-14) continue;
                double w0 = (bt[i] * at1[j] - at1[i] * bt[j]) / det;
                double w1 = (at0[i] * bt[j] - bt[i] * at0[j]) / det;
                double c = w0 * w0 + w1 * w1;
                if (c < best && orc_feasible(M, at0, at1, bt, w0, w1, tol)) {
                    best = c; bw0 = w0; bw1 = w1;
                }
            }
        }
    }
    if (!(best < INFINITY)) return 1;
    double w[2] = {bw0, bw1};
    orc_from_wspace(w, u_nom, l00, l10, l11, u);
    return 0;
}

/* Second, independent algorithm for the same problem (cross-check of
 * orc_qp2_enum): rows are inserted one at a time; when the current optimum
 * violates the new row, the optimum moves onto that row's line and the
 * 1-D problem along it, bounded by the earlier rows, is solved. */
int orc_qp2_incremental(int M, const double* rows, const double R[4], const double u_nom[2],
                        double u[2], double* work)
{
    double l00, l10, l11;
    u[0] = u_nom[0];
    u[1] = u_nom[1];
    if (!orc_chol2(R, &l00, &l10, &l11)) return 1;
    double *at0 = work, *at1 = work + M, *bt = work + 2 * M;
    orc_to_wspace(M, rows, u_nom, l00, l10, l11, at0, at1, bt);
    const double tol = 1e-12;
    double w0 = 0.0, w1 = 0.0;
    for (int i = 0; i < M; ++i) {
        if (at0[i] == 0.0 && at1[i] == 0.0) {
            if (bt[i] > 0.0) return 1;
            continue;
        }
        double lhs = at0[i] * w0 + at1[i] * w1;
        double scale = fabs(at0[i] * w0) + fabs(at1[i] * w1) + fabs(bt[i]);
        if (lhs - bt[i] >= -tol * scale) continue;
        double p0 = at0[i] * bt[i], p1 = at1[i] * bt[i];
        double d0 = -at1[i], d1 = at0[i];
        double tlo = -INFINITY, thi = INFINITY;
        for (int j = 0; j < i; ++j) {
            if (at0[j] == 0.0 && at1[j] == 0.0) continue;
            double c = at0[j] * d0 + at1[j] * d1;
            double e = bt[j] - (at0[j] * p0 + at1[j] * p1);
            if (fabs(c) > 1e-14) {
                double t = e / c;
                if (c > 0.0) { if (t > tlo) tlo = t; }
                else { if (t < thi) thi = t; }
            } else if (e > 1e-10 * (fabs(bt[j]) + fabs(bt[i]) + 1e-300)) {
                return 1;
            }
        }
        if (tlo > thi + 1e-10 * (fabs(tlo) + fabs(thi) + 1e-300)) return 1;
        double t = 0.0;
        if (t < tlo) t = tlo;
        if (t > thi) t = thi;
        w0 = p0 + t * d0;
        w1 = p1 + t * d1;
    }
    double w[2] = {w0, w1};
    orc_from_wspace(w, u_nom, l00, l10, l11, u);
    return 0;
}

/* Rows that admit no input: the reference hands cvxopt an infeasible program and applies whatever comes back
 * (cbf/cbf.py:213-220, the status is never read).  Defined here as the safest input there is: the one whose largest
 * row violation, measured as the distance to the row's half-plane in the metric of the cost (unit normals in w), is
 * least; among those the one nearest to the nominal input.  Exact, by enumeration: the least worst-case violation t*
 * of the linear program  min t  s.t.  at_i . w + t >= bt_i  is the largest value any dual vertex gives, and a dual
 * vertex is an antiparallel pair (t = (bt_i + bt_j) / 2) or a triple whose normals span the plane positively
 * (t = sum D_i bt_i / sum D_i with D the cross products of the other two normals); then the QP on the rows relaxed by
 * t*.  Rows with a zero normal cannot be helped by any input and are left out.  returns 0, t_out = t* (<= 0: the
 * rows were feasible and the ordinary optimum is returned). */
int orc_qp2_min_violation(int M, const double* rows, const double R[4], const double u_nom[2],
                          double u[2], double* t_out, double* work /* 6*M doubles */)
{
    double l00, l10, l11;
    u[0] = u_nom[0];
    u[1] = u_nom[1];
    *t_out = 0.0;
    if (!orc_chol2(R, &l00, &l10, &l11)) return 1;
    double *at0 = work, *at1 = work + M, *bt = work + 2 * M, *relaxed = work + 3 * M;
    orc_to_wspace(M, rows, u_nom, l00, l10, l11, at0, at1, bt);
    double tstar = -INFINITY;
    for (int i = 0; i < M; ++i) {
        if (at0[i] == 0.0 && at1[i] == 0.0) continue;
        for (int j = i + 1; j < M; ++j) {
            if (at0[j] == 0.0 && at1[j] == 0.0) continue;
            double cij = at0[i] * at1[j] - at1[i] * at0[j];
            double dij = at0[i] * at0[j] + at1[i] * at1[j];
            if (fabs(cij) <= 1e-14 && dij < 0.0) {
                double t = 0.5 * (bt[i] + bt[j]);
                if (t > tstar) tstar = t;
            }
            for (int k = j + 1; k < M; ++k) {
                if (at0[k] == 0.0 && at1[k] == 0.0) continue;
                double d1 = at0[j] * at1[k] - at1[j] * at0[k];
                double d2 = at0[k] * at1[i] - at1[k] * at0[i];
                double d3 = cij;
                if (!((d1 > 1e-14 && d2 > 1e-14 && d3 > 1e-14) || (d1 < -1e-14 && d2 < -1e-14 && d3 < -1e-14))) continue;
                double t = (d1 * bt[i] + d2 * bt[j] + d3 * bt[k]) / (d1 + d2 + d3);
                if (t > tstar) tstar = t;
            }
        }
    }
    *t_out = tstar;
    double relax = tstar > 0.0 ? tstar * (1.0 + 1e-12) : 0.0;
    int n = 0;
    for (int i = 0; i < M; ++i) {
        double a0 = rows[3 * i], a1 = rows[3 * i + 1], b = rows[3 * i + 2];
        if (at0[i] == 0.0 && at1[i] == 0.0) continue;   /* left out */
        double t0 = a0 / l00, t1 = (a1 - l10 * t0) / l11;
        double nrm = sqrt(t0 * t0 + t1 * t1);
        relaxed[3 * n] = a0;
        relaxed[3 * n + 1] = a1;
        relaxed[3 * n + 2] = b - relax * nrm;
        ++n;
    }
    return orc_qp2_enum(n, relaxed, R, u_nom, u, work);   /* work[0 .. 3n) is free again: its rows sit behind it */
}

int orc_qp_batch(const orc_params* p, int64_t A, int32_t M, const double* rows_in, const double* u_nom_in,
                 double* u_out, int32_t* status, int use_incremental);

/* orc_qp_batch with the least-violation answer for the agents whose rows admit no input: u_out holds it, status
 * stays 1, t_out[a] = the least worst-case violation (0 for the others). */
int orc_qp_min_violation_batch(const orc_params* p, int64_t A, int32_t M, const double* rows_in,
                               const double* u_nom_in, double* u_out, int32_t* status, double* t_out)
{
    if (!p || A < 0 || M < 0 || M > 64) return -1;   /* enumeration over triples: small M only */
    int rc = orc_qp_batch(p, A, M, rows_in, u_nom_in, u_out, status, 0);
    if (rc != 0) return rc;
#pragma omp parallel for schedule(static)
    for (int64_t a = 0; a < A; ++a) {
        t_out[a] = 0.0;
        if (status[a] != 1) continue;
        double rows[3 * 64], work[6 * 64];
        for (int m = 0; m < M; ++m) {
            const double* ri = rows_in + ((size_t)m * A + a) * 4;
            rows[3 * m] = ri[0]; rows[3 * m + 1] = ri[1]; rows[3 * m + 2] = ri[2];
        }
        double u_nom[2] = {u_nom_in[2 * a], u_nom_in[2 * a + 1]}, u[2], t;
        if (orc_qp2_min_violation(M, rows, p->R, u_nom, u, &t, work) == 0) {
            u_out[2 * a] = u[0];
            u_out[2 * a + 1] = u[1];
            t_out[a] = t > 0.0 ? t : 0.0;   /* only rows that no input satisfies stood in the way: nothing left to violate */
        }
    }
    return 0;
}

/* ------------------------------------------------------------ whole tick, batched */

static int orc_finite(double x) { return isfinite(x); }

/*
 * One control tick for A agents; same data layout as sccbf_solve_f64 in
 * include/sccbf.h but with HOST pointers.
 * DBM path: DBM_CBF_2DS.solve_cbf, cbf/cbf.py:166-220 (delta->beta 175,
 * beta->delta 216).  KBM path: KBM_VC_CBF2D.solve_cbf, cbf/cbf.py:67-110
 * (omega_ref 75, delta 109).
 * use_incremental selects the cross-check QP algorithm.
 */
int orc_solve_batch(const orc_params* p, int64_t A, int32_t M, const double* state,
                    const double* u_ref, const double* obs, const uint8_t* obs_type,
                    double* u_out, int32_t* status, double* rows_out, int use_incremental)
{
    if (!p || A < 0 || M < 0 || M > 1024) return -1;
    const int shared = (p->flags & ORC_FLAG_OBS_SHARED) != 0;
    const int type_pa = (p->flags & ORC_FLAG_TYPE_PER_AGENT) != 0;
#pragma omp parallel for schedule(static)
    for (int64_t a = 0; a < A; ++a) {
        double rows[3 * 1024 > 0 ? 3 * 1024 : 1];
        double work[3 * 1024];
        const double* s = state + 4 * a;
        double ur0 = u_ref[2 * a], ur1 = u_ref[2 * a + 1];
        int bad = !(orc_finite(s[0]) && orc_finite(s[1]) && orc_finite(s[2]) && orc_finite(s[3]) &&
                    orc_finite(ur0) && orc_finite(ur1));
        double u_nom[2];
        u_nom[0] = ur0;
        if ((p->flags & ORC_FLAG_SLIP_IO) || p->dynamics == ORC_DYN_DUM)
            u_nom[1] = ur1;
        else if (p->dynamics == ORC_DYN_KBM)
            u_nom[1] = ur0 * tan(ur1) / p->L;
        else
            u_nom[1] = atan2(p->lr * tan(ur1), p->lf + p->lr);
        for (int m = 0; m < M; ++m) {
            double o[8];
            for (int h = 0; h < 2; ++h)
                for (int c = 0; c < 4; ++c)
                    o[4 * h + c] = shared ? obs[((size_t)m * 2 + h) * 4 + c]
                                          : obs[(((size_t)m * 2 + h) * A + a) * 4 + c];
            int type = type_pa ? obs_type[(size_t)m * A + a] : obs_type[m];
            orc_partials d;
            double* row = rows + 3 * m;
            if (orc_barrier(type, s, o, p->zero_tol, &d)) {
                orc_row(p, s, &d, row);
                if (!(orc_finite(row[0]) && orc_finite(row[1]) && orc_finite(row[2]))) bad = 1;
            } else {
                row[0] = 0.0; row[1] = 0.0; row[2] = -1.0;
            }
            if (rows_out) {
                double* ro = rows_out + ((size_t)m * A + a) * 4;
                ro[0] = row[0]; ro[1] = row[1]; ro[2] = row[2]; ro[3] = d.h;
            }
        }
        double u[2];
        int st;
        if (bad) {
            st = 2;
        } else {
            st = use_incremental ? orc_qp2_incremental(M, rows, p->R, u_nom, u, work)
                                 : orc_qp2_enum(M, rows, p->R, u_nom, u, work);
        }
        if (st != 0) {
            u_out[2 * a] = ur0;
            u_out[2 * a + 1] = ur1;
        } else if ((p->flags & ORC_FLAG_SLIP_IO) || p->dynamics == ORC_DYN_DUM) {
            u_out[2 * a] = u[0];
            u_out[2 * a + 1] = u[1];
        } else if (p->dynamics == ORC_DYN_KBM) {
            u_out[2 * a] = u[0];
            u_out[2 * a + 1] = atan2(u[1] * p->L, ur0);
        } else {
            u_out[2 * a] = u[0];
            u_out[2 * a + 1] = atan2((p->lf + p->lr) * tan(u[1]), p->lr);
        }
        status[a] = st;
    }
    return 0;
}

/*
 * SADBM_CBF_2DS.solve_cbf, cbf/cbf.py:349-437, for A agents with a given time step (the
 * reference's fixed-dt mode, 331-332).  slip is [A][2] in/out = beta, beta_ref_last.
 * delta_ref -> beta_ref (359), beta_ref_dot (367), nominal input [a_ref, beta_ref_dot] (372),
 * the QP, beta += beta_dot dt (421), delta = atan2((lf + lr) tan(beta), lr) (430).  When the QP
 * has no answer the nominal input is integrated instead; non-finite data pass through.
 * With ORC_FLAG_CONE_BETA_STATE the cones take the agent's beta (what cbf.py:424-427 leaves in
 * them for the next call), else the beta of their slot.
 */
int orc_solve_sa_batch(const orc_params* p, double dt, int64_t A, int32_t M, const double* state,
                       double* slip, const double* u_ref, const double* obs, const uint8_t* obs_type,
                       double* u_out, int32_t* status, double* rows_out, int use_incremental)
{
    if (!p || A < 0 || M < 1 || M > 1024 || !(dt > 0.0)) return -1;
    const int shared = (p->flags & ORC_FLAG_OBS_SHARED) != 0;
    const int type_pa = (p->flags & ORC_FLAG_TYPE_PER_AGENT) != 0;
#pragma omp parallel for schedule(static)
    for (int64_t a = 0; a < A; ++a) {
        double rows[3 * 1024];
        double work[3 * 1024];
        const double* s = state + 4 * a;
        double ur0 = u_ref[2 * a], ur1 = u_ref[2 * a + 1];
        double beta = slip[2 * a], beta_ref_last = slip[2 * a + 1];
        int bad = !(orc_finite(s[0]) && orc_finite(s[1]) && orc_finite(s[2]) && orc_finite(s[3]) &&
                    orc_finite(ur0) && orc_finite(ur1) && orc_finite(beta) && orc_finite(beta_ref_last));
        double beta_ref = atan2(p->lr * tan(ur1), p->lf + p->lr);
        double u_nom[2] = {ur0, (beta_ref - beta_ref_last) / dt};
        for (int m = 0; m < M; ++m) {
            double o[8];
            for (int h = 0; h < 2; ++h)
                for (int c = 0; c < 4; ++c)
                    o[4 * h + c] = shared ? obs[((size_t)m * 2 + h) * 4 + c]
                                          : obs[(((size_t)m * 2 + h) * A + a) * 4 + c];
            int type = type_pa ? obs_type[(size_t)m * A + a] : obs_type[m];
            orc_unbox(&type, o);
            if (type == ORC_OBS_CONE && (p->flags & ORC_FLAG_CONE_BETA_STATE)) o[5] = beta;
            orc_partials d;
            double* row = rows + 3 * m;
            if (orc_barrier(type, s, o, p->zero_tol, &d)) {
                orc_row_sa(p, s, beta, &d, (type == ORC_OBS_CONE || type == ORC_OBS_CONE_PREP) ? d.hth : 0.0, row);
                if (!(orc_finite(row[0]) && orc_finite(row[1]) && orc_finite(row[2]))) bad = 1;
            } else {
                row[0] = 0.0; row[1] = 0.0; row[2] = -1.0;
            }
            if (rows_out) {
                double* ro = rows_out + ((size_t)m * A + a) * 4;
                ro[0] = row[0]; ro[1] = row[1]; ro[2] = row[2]; ro[3] = d.h;
            }
        }
        double u[2] = {u_nom[0], u_nom[1]};
        int st = 2;
        if (!bad) {
            st = use_incremental ? orc_qp2_incremental(M, rows, p->R, u_nom, u, work)
                                 : orc_qp2_enum(M, rows, p->R, u_nom, u, work);
            if (st != 0) { u[0] = u_nom[0]; u[1] = u_nom[1]; }
        }
        status[a] = st;
        if (st == 2) {
            u_out[2 * a] = ur0;
            u_out[2 * a + 1] = ur1;
            continue;
        }
        beta += u[1] * dt;
        slip[2 * a] = beta;
        slip[2 * a + 1] = beta_ref;
        u_out[2 * a] = u[0];
        u_out[2 * a + 1] = atan2((p->lf + p->lr) * tan(beta), p->lr);
    }
    return 0;
}

/* The QP stage alone on caller-supplied rows T[M][A][4] (4th scalar ignored);
 * counterpart of sccbf_qp_f64. */
int orc_qp_batch(const orc_params* p, int64_t A, int32_t M, const double* rows_in,
                 const double* u_nom_in, double* u_out, int32_t* status, int use_incremental)
{
    if (!p || A < 0 || M < 0 || M > 1024) return -1;
#pragma omp parallel for schedule(static)
    for (int64_t a = 0; a < A; ++a) {
        double rows[3 * 1024];
        double work[3 * 1024];
        int bad = 0;
        for (int m = 0; m < M; ++m) {
            const double* ri = rows_in + ((size_t)m * A + a) * 4;
            rows[3 * m] = ri[0]; rows[3 * m + 1] = ri[1]; rows[3 * m + 2] = ri[2];
            if (!(orc_finite(ri[0]) && orc_finite(ri[1]) && orc_finite(ri[2]))) bad = 1;
        }
        double u_nom[2] = {u_nom_in[2 * a], u_nom_in[2 * a + 1]};
        if (!(orc_finite(u_nom[0]) && orc_finite(u_nom[1]))) bad = 1;
        double u[2] = {u_nom[0], u_nom[1]};
        int st = 2;
        if (!bad)
            st = use_incremental ? orc_qp2_incremental(M, rows, p->R, u_nom, u, work)
                                 : orc_qp2_enum(M, rows, p->R, u_nom, u, work);
        if (st != 0) { u[0] = u_nom[0]; u[1] = u_nom[1]; }
        u_out[2 * a] = u[0];
        u_out[2 * a + 1] = u[1];
        status[a] = st;
    }
    return 0;
}

/* ------------------------------------------------- callers either side of the path */

/* State.update_com, test_scripts/stanley_controller_ellipse.py:122-131.
 * s = x, y, yaw, v (updated in place); returns the slip angle beta. */
double orc_vehicle_step(double s[4], double acc, double delta, double lr, double lf,
                        double max_steer, double dt)
{
    if (delta > max_steer) delta = max_steer;
    if (delta < -max_steer) delta = -max_steer;
    double beta = atan2(lr * tan(delta), lf + lr);
    double x = s[0], y = s[1], th = s[2], v = s[3];
    s[0] = x + (v * cos(th) - v * sin(th) * beta) * dt;
    s[1] = y + (v * sin(th) + v * cos(th) * beta) * dt;
    s[2] = th + (v * beta / lr) * dt;
    s[3] = v + acc * dt;
    return beta;
}

static double orc_normalize_angle(double a)
{
    /* cbf/utils.py:93-106 subtracts / adds 2 pi until the angle is in [-pi, pi]; an angle of
     * many turns (a diverged state) would keep that loop busy for hours and an infinite one for
     * ever: such angles are brought near first, non-finite ones returned as they are */
    if (!isfinite(a)) return a;
    if (fabs(a) > 64.0) a = remainder(a, 2.0 * M_PI);
    while (a > M_PI) a -= 2.0 * M_PI;
    while (a < -M_PI) a += 2.0 * M_PI;
    return a;
}

/* Stanley steering as the driver script states it,
 * test_scripts/stanley_controller_ellipse.py:146-169 and 188-212: nearest
 * course point to the front axle (offset `axle` along the heading), index kept
 * monotone, delta = heading error + atan2(k e, v + ks).
 * (LateralStanley.control, cbf/controllers.py:69-151, is the same law with
 * axle = lf and softening ks.)  Returns delta, writes the target index. */
double orc_stanley(const double s[4], int n, const double* cx, const double* cy,
                   const double* cyaw, int last_idx, double axle, double k, double ks,
                   int* idx_out)
{
    double fx = s[0] + axle * cos(s[2]);
    double fy = s[1] + axle * sin(s[2]);
    int best = 0;
    double bd = INFINITY;
    for (int i = 0; i < n; ++i) {
        double d = hypot(fx - cx[i], fy - cy[i]);
        if (d < bd) { bd = d; best = i; }
    }
    double ex = -cos(s[2] + M_PI / 2.0), ey = -sin(s[2] + M_PI / 2.0);
    double err = (fx - cx[best]) * ex + (fy - cy[best]) * ey;
    int idx = best;
    if (last_idx >= idx) idx = last_idx;
    /* a caller's index beyond the course (the reference would raise IndexError) reads the last way-point */
    double theta_e = orc_normalize_angle(cyaw[idx < n ? idx : n - 1] - s[2]);
    double theta_d = atan2(k * err, s[3] + ks);
    *idx_out = idx;
    return theta_e + theta_d;
}

/* Nominal input for a batch of agents on one shared course: Stanley steering as above and
 * the scalar speed controller PID1.control, cbf/controllers.py:173-180 (e = target - v,
 * de = (e - e_prev) / dt, ie += dt e, a = kp e + ki ie + kd de).
 * traj is [N][4] = x, y, yaw, v_des; target_idx [A] in/out; pid_state [A][2] = e_prev, ie
 * in/out or NULL (then a = kp e); u_ref [A][2] = a_ref, delta_ref. */
void orc_nominal_batch(int64_t A, int32_t N, const double* state, const double* traj,
                       int32_t* target_idx, double* pid_state, double axle, double k, double ks,
                       double kp, double ki, double kd, double dt, double target_speed,
                       double* u_ref)
{
    double* cx = (double*)malloc(sizeof(double) * 3 * (size_t)(N > 0 ? N : 1));
    double* cy = cx + N;
    double* cyaw = cy + N;
    for (int i = 0; i < N; ++i) {
        cx[i] = traj[4 * i];
        cy[i] = traj[4 * i + 1];
        cyaw[i] = traj[4 * i + 2];
    }
#pragma omp parallel for schedule(static)
    for (int64_t a = 0; a < A; ++a) {
        int idx = 0;
        double delta = orc_stanley(state + 4 * a, N, cx, cy, cyaw, target_idx[a], axle, k, ks, &idx);
        double e = target_speed - state[4 * a + 3];
        double acc = kp * e;
        if (pid_state) {
            double de = (e - pid_state[2 * a]) / dt;
            pid_state[2 * a + 1] += dt * e;
            acc = kp * e + ki * pid_state[2 * a + 1] + kd * de;
            pid_state[2 * a] = e;
        }
        target_idx[a] = idx;
        u_ref[2 * a] = acc;
        u_ref[2 * a + 1] = delta;
    }
    free(cx);
}

/* orc_vehicle_step over a batch: state [A][4] in place, u [A][2] = a, delta; beta [A] or NULL. */
void orc_vehicle_step_batch(int64_t A, double* state, const double* u, double lr, double lf,
                            double max_steer, double dt, double* beta)
{
#pragma omp parallel for schedule(static)
    for (int64_t a = 0; a < A; ++a) {
        double b = orc_vehicle_step(state + 4 * a, u[2 * a], u[2 * a + 1], lr, lf, max_steer, dt);
        if (beta) beta[a] = b;
    }
}
