/*
 * ba_oracle.c — CPU restatement of the bundle-adjustment path (see ba_oracle.h for status).
 * TEST INFRASTRUCTURE ONLY; never linked into the product.  All arithmetic in double like g2o.
 */
#include "ba_oracle.h"
#include "orb_oracle.h" /* orc_get_convention: tools/convention_sensitivity.py */

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---------------- SE3Quat (code/Thirdparty/g2o/g2o/types/se3quat.h) ---------------- */
typedef struct { double q[4]; /* x y z w */ double t[3]; } se3;

static void quat_normalize_rotation(double* q) { /* SE3Quat::normalizeRotation, se3quat.h:269-274 */
    if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

/* Eigen::Quaterniond(Matrix3d) — published algorithm of Eigen/src/Geometry/Quaternion.h
 * (quaternionbase_assign_impl<Other,3,3>), restated; R row-major. */
static void quat_from_R(const double* R, double* q) {
    double t = R[0] + R[4] + R[8];
    if (orc_get_convention() & ORC_CONV_QUAT_LARGEST) {
        /* alternative: Shepperd's method with the largest of (trace, R00, R11, R22) as the pivot */
        int best = 3;
        double bv = t;
        for (int i = 0; i < 3; i++)
            if (R[i * 3 + i] > bv) { bv = R[i * 3 + i]; best = i; }
        if (best == 3) {
            const double w4 = 2.0 * sqrt(t + 1.0);
            q[3] = 0.25 * w4;
            q[0] = (R[7] - R[5]) / w4; q[1] = (R[2] - R[6]) / w4; q[2] = (R[3] - R[1]) / w4;
        } else {
            const int i = best, j = (i + 1) % 3, k = (j + 1) % 3;
            const double s4 = 2.0 * sqrt(1.0 + R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k]);
            q[i] = 0.25 * s4;
            q[3] = (R[k * 3 + j] - R[j * 3 + k]) / s4;
            q[j] = (R[j * 3 + i] + R[i * 3 + j]) / s4;
            q[k] = (R[k * 3 + i] + R[i * 3 + k]) / s4;
        }
        return;
    }
    if (t > 0.0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (R[7] - R[5]) * t;
        q[1] = (R[2] - R[6]) * t;
        q[2] = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[i * 3 + i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (R[k * 3 + j] - R[j * 3 + k]) * t;
        q[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
        q[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
    }
}

/* Eigen QuaternionBase::toRotationMatrix */
static void quat_to_R(const double* q, double* R) {
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

/* Eigen QuaternionBase::_transformVector: v + w*(2 u x v) + u x (2 u x v) */
static void quat_rotate(const double* q, const double* v, double* out) {
    double uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    out[0] = v[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
    out[1] = v[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
    out[2] = v[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}

static void quat_mul(const double* a, const double* b, double* o) {
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
}

/* Converter::toSE3Quat (code/src/Converter.cc:37-47) -> SE3Quat(R,t) (se3quat.h:58-60) */
void orc_se3_from_Tcw(const float* T, double* q, double* t) {
    const double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    quat_from_R(R, q);
    quat_normalize_rotation(q);
    t[0] = T[3]; t[1] = T[7]; t[2] = T[11];
}

/* Converter::toCvMat(SE3Quat) (Converter.cc:49-53,66-74): to_homogeneous_matrix cast to float */
void orc_se3_to_Tcw(const double* q, const double* t, float* T) {
    double R[9];
    quat_to_R(q, R);
    T[0] = (float)R[0]; T[1] = (float)R[1]; T[2] = (float)R[2];  T[3] = (float)t[0];
    T[4] = (float)R[3]; T[5] = (float)R[4]; T[6] = (float)R[5];  T[7] = (float)t[1];
    T[8] = (float)R[6]; T[9] = (float)R[7]; T[10] = (float)R[8]; T[11] = (float)t[2];
}

static void mat3_mul(const double* A, const double* B, double* C) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}

/* VertexSE3Expmap::oplusImpl: estimate <- SE3Quat::exp(update) * estimate
 * (types_six_dof_expmap.h:73-76; se3quat.h:223-257 exp; :104-110 operator*) */
void orc_se3_exp_mul(const double* u, double* q, double* t) {
    const double omega[3] = {u[0], u[1], u[2]}, upsilon[3] = {u[3], u[4], u[5]};
    const double theta = sqrt(omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2]);
    const double Om[9] = {0, -omega[2], omega[1], omega[2], 0, -omega[0], -omega[1], omega[0], 0};
    double Om2[9], R[9], V[9];
    mat3_mul(Om, Om, Om2);
    if (theta < 0.00001) {
        for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0 ? 1.0 : 0.0) + Om[i] + Om2[i];
        memcpy(V, R, sizeof(R));
    } else {
        const double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta);
        const double c = (theta - sin(theta)) / pow(theta, 3);
        for (int i = 0; i < 9; i++) {
            R[i] = (i % 4 == 0 ? 1.0 : 0.0) + a * Om[i] + b * Om2[i];
            V[i] = (i % 4 == 0 ? 1.0 : 0.0) + b * Om[i] + c * Om2[i];
        }
    }
    double qa[4], ta[3];
    quat_from_R(R, qa);
    quat_normalize_rotation(qa); /* SE3Quat(Quaterniond, t) ctor normalises */
    for (int i = 0; i < 3; i++) ta[i] = V[i * 3] * upsilon[0] + V[i * 3 + 1] * upsilon[1] + V[i * 3 + 2] * upsilon[2];
    double rt[3], qn[4];
    quat_rotate(qa, t, rt);
    t[0] = ta[0] + rt[0]; t[1] = ta[1] + rt[1]; t[2] = ta[2] + rt[2];
    quat_mul(qa, q, qn);
    memcpy(q, qn, sizeof(qn));
    quat_normalize_rotation(q);
}

/* EdgeSE3ProjectXYZ::computeError (types_six_dof_expmap.h:89-94) + linearizeOplus (.cpp:103-138) */
double orc_edge_project(const double* q, const double* t, const double* X, const double* obs, const double* intr,
                        double* err, double* Jp, double* Jc) {
    const double fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
    double pc[3];
    quat_rotate(q, X, pc);
    pc[0] += t[0]; pc[1] += t[1]; pc[2] += t[2];
    const double x = pc[0], y = pc[1], z = pc[2];
    if (err) {
        err[0] = obs[0] - (x / z * fx + cx); /* project2d then cam_project */
        err[1] = obs[1] - (y / z * fy + cy);
    }
    if (Jp) {
        double R[9];
        quat_to_R(q, R);
        const double z_2 = z * z;
        const double tmp[6] = {fx, 0, -x / z * fx, 0, fy, -y / z * fy};
        for (int r = 0; r < 2; r++)
            for (int c = 0; c < 3; c++)
                Jp[r * 3 + c] = -1. / z * (tmp[r * 3] * R[c] + tmp[r * 3 + 1] * R[3 + c] + tmp[r * 3 + 2] * R[6 + c]);
        Jc[0] = x * y / z_2 * fx;  Jc[1] = -(1 + (x * x / z_2)) * fx; Jc[2] = y / z * fx;
        Jc[3] = -1. / z * fx;      Jc[4] = 0;                        Jc[5] = x / z_2 * fx;
        Jc[6] = (1 + y * y / z_2) * fy; Jc[7] = -x * y / z_2 * fy;   Jc[8] = -x / z * fy;
        Jc[9] = 0;                 Jc[10] = -1. / z * fy;            Jc[11] = y / z_2 * fy;
    }
    return z;
}

/* RobustKernelHuber::robustify (robust_kernel_impl.cpp:78-91); dsqr is stored as float (impl.h:84) */
void orc_huber(double e, double delta, double* rho) {
    const float dsqr = (float)(delta * delta);
    if (e <= dsqr) {
        rho[0] = e; rho[1] = 1.; rho[2] = 0.;
    } else {
        const double sqrte = sqrt(e);
        rho[0] = 2 * sqrte * delta - dsqr;
        rho[1] = delta / sqrte;
        rho[2] = -0.5 * rho[1] / e;
    }
}

/* ---------------- the optimiser ---------------- */
typedef struct {
    const orc_ba_problem* p;
    se3* pose; double (*pt)[3];
    se3* pose_bak; double (*pt_bak)[3];
    double (*intr)[4];
    /* per edge */
    int* level; int robust; double delta;
    double (*err)[2];       /* stored _error */
    /* active structure of the current stage */
    int* pose_hidx; int* pt_hidx; int np, nl; /* hessian indices (-1 = not optimised) */
    int* active; int n_active;                 /* active edge list, insertion order */
    double* Hpp;  /* np x 36 */
    double* Hll;  /* nl x 9 */
    double* Hpl;  /* per active edge 6x3 (pose x point), zero if pose fixed */
    double* b;    /* 6 np + 3 nl */
    double* x;
    double lambda, ni; int nBad;
    const volatile uint8_t* stop;
    int trials;
} ba;

static int terminate(const ba* s) { return s->stop && *s->stop; }

static void init_stage(ba* s) { /* SparseOptimizer::initializeOptimization(0) (sparse_optimizer.cpp:196-270) */
    const orc_ba_problem* p = s->p;
    for (int i = 0; i < p->n_poses; i++) s->pose_hidx[i] = -1;
    for (int i = 0; i < p->n_points; i++) s->pt_hidx[i] = -1;
    s->n_active = 0;
    for (int e = 0; e < p->n_edges; e++)
        if (s->level[e] == 0) { /* the point is never fixed, so !allVerticesFixed() always holds */
            s->active[s->n_active++] = e;
            s->pose_hidx[p->edge_pose[e]] = -2; /* touched */
            s->pt_hidx[p->edge_point[e]] = -2;
        }
    s->np = s->nl = 0; /* buildIndexMapping: non-marginalised (poses) first, then landmarks, by id */
    for (int i = 0; i < p->n_poses; i++)
        s->pose_hidx[i] = (s->pose_hidx[i] == -2 && !p->fixed[i]) ? s->np++ : -1;
    for (int i = 0; i < p->n_points; i++) s->pt_hidx[i] = (s->pt_hidx[i] == -2) ? s->nl++ : -1;
}

static void compute_active_errors(ba* s) {
    const orc_ba_problem* p = s->p;
    for (int k = 0; k < s->n_active; k++) {
        const int e = s->active[k];
        const double obs[2] = {p->obs[2 * e], p->obs[2 * e + 1]};
        const int ip = p->edge_pose[e];
        orc_edge_project(s->pose[ip].q, s->pose[ip].t, s->pt[p->edge_point[e]], obs, s->intr[ip], s->err[e], 0, 0);
    }
}

static double edge_chi2(const ba* s, int e) { /* _error.dot(information()*_error) */
    const double w = (double)s->p->inv_sigma2[e];
    return s->err[e][0] * (w * s->err[e][0]) + s->err[e][1] * (w * s->err[e][1]);
}

static double active_robust_chi2(const ba* s) {
    double chi = 0.0, rho[3];
    for (int k = 0; k < s->n_active; k++) {
        const int e = s->active[k];
        if (s->robust) {
            orc_huber(edge_chi2(s, e), s->delta, rho);
            chi += rho[0];
        } else chi += edge_chi2(s, e);
    }
    return chi;
}

/* BlockSolver::buildSystem (block_solver.hpp:502-560) + constructQuadraticForm (base_binary_edge.hpp:55-120) */
static void build_system(ba* s) {
    const orc_ba_problem* p = s->p;
    memset(s->Hpp, 0, sizeof(double) * 36 * (size_t)s->np);
    memset(s->Hll, 0, sizeof(double) * 9 * (size_t)s->nl);
    memset(s->Hpl, 0, sizeof(double) * 18 * (size_t)s->n_active);
    memset(s->b, 0, sizeof(double) * (size_t)(6 * s->np + 3 * s->nl));
    for (int k = 0; k < s->n_active; k++) {
        const int e = s->active[k];
        const int ip = p->edge_pose[e], il = p->edge_point[e];
        const double obs[2] = {p->obs[2 * e], p->obs[2 * e + 1]};
        double er[2], A[6], B[12];
        orc_edge_project(s->pose[ip].q, s->pose[ip].t, s->pt[il], obs, s->intr[ip], er, A, B);
        /* _error was set by the preceding computeActiveErrors with the same estimates */
        const double om = (double)p->inv_sigma2[e];
        double w = om, orr[2] = {-om * s->err[e][0], -om * s->err[e][1]};
        if (s->robust) {
            double rho[3];
            orc_huber(edge_chi2(s, e), s->delta, rho);
            w = rho[1] * om; /* robustInformation */
            orr[0] *= rho[1];
            orr[1] *= rho[1];
        }
        const int hl = s->pt_hidx[il], hp = s->pose_hidx[ip];
        double* bl = s->b + 6 * s->np + 3 * hl;
        double* Hl = s->Hll + 9 * (size_t)hl;
        for (int r = 0; r < 3; r++) {
            bl[r] += A[r] * orr[0] + A[3 + r] * orr[1];
            for (int c = 0; c < 3; c++) Hl[r * 3 + c] += A[r] * w * A[c] + A[3 + r] * w * A[3 + c];
        }
        if (hp >= 0) {
            double* bp = s->b + 6 * hp;
            double* Hp = s->Hpp + 36 * (size_t)hp;
            double* W = s->Hpl + 18 * (size_t)k; /* pose x point = B^T w A */
            for (int r = 0; r < 6; r++) {
                bp[r] += B[r] * orr[0] + B[6 + r] * orr[1];
                for (int c = 0; c < 6; c++) Hp[r * 6 + c] += B[r] * w * B[c] + B[6 + r] * w * B[6 + c];
                for (int c = 0; c < 3; c++) W[r * 3 + c] += B[r] * w * A[c] + B[6 + r] * w * A[3 + c];
            }
        }
        (void)er;
    }
}

/* alternative (ORC_CONV_LDLT_REVERSED): S = P^T L D L^T P with P the reversal of the unknowns, no square roots - another
 * elimination order and another factorisation than the Cholesky below, the kind of difference a fill-reducing
 * SimplicialLDLT has against it */
static int ldlt_reversed_solve(double* S, double* rhs, int n) {
    double* A = (double*)malloc(sizeof(double) * (size_t)n * n);
    double* b = (double*)malloc(sizeof(double) * (size_t)n);
    for (int i = 0; i < n; i++) {
        b[i] = rhs[n - 1 - i];
        for (int j = 0; j < n; j++) {  /* the caller fills the lower triangle (row >= col) */
            const int r = n - 1 - i, c = n - 1 - j;
            A[i * (size_t)n + j] = r >= c ? S[r * (size_t)n + c] : S[c * (size_t)n + r];
        }
    }
    int ok = 1;
    for (int j = 0; j < n && ok; j++) {  /* A = L D L^T, L unit lower in the strict lower triangle, D on the diagonal */
        double d = A[j * (size_t)n + j];
        for (int k = 0; k < j; k++) d -= A[j * (size_t)n + k] * A[j * (size_t)n + k] * A[k * (size_t)n + k];
        if (!(d > 0.0)) { ok = 0; break; }
        A[j * (size_t)n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double v = A[i * (size_t)n + j];
            for (int k = 0; k < j; k++) v -= A[i * (size_t)n + k] * A[j * (size_t)n + k] * A[k * (size_t)n + k];
            A[i * (size_t)n + j] = v / d;
        }
    }
    if (ok) {
        for (int i = 0; i < n; i++) { double v = b[i]; for (int k = 0; k < i; k++) v -= A[i * (size_t)n + k] * b[k]; b[i] = v; }
        for (int i = 0; i < n; i++) b[i] /= A[i * (size_t)n + i];
        for (int i = n - 1; i >= 0; i--) { double v = b[i]; for (int k = i + 1; k < n; k++) v -= A[k * (size_t)n + i] * b[k]; b[i] = v; }
        for (int i = 0; i < n; i++) rhs[n - 1 - i] = b[i];
    }
    free(A); free(b);
    return ok;
}

static int cholesky_solve(double* S, double* rhs, int n) { /* in place; returns 0 on a non-positive pivot */
    if (orc_get_convention() & ORC_CONV_LDLT_REVERSED) return ldlt_reversed_solve(S, rhs, n);
    /* first[i] = first structurally nonzero column of row i.  Cholesky fill stays inside this row envelope, so every
     * product the loops below skip has an exact zero factor: the result is bit-identical to the full loops (what
     * LinearSolverEigen's SimplicialLDLT does with its elimination tree, linear_solver_eigen.h:147-232), and a merged
     * multi-agent map (block-banded, GBA-2r) costs a fifth of the dense count.  (tools/make_gba_golden.py) */
    int* first = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++) {
        int f = 0;
        while (f < i && S[i * (size_t)n + f] == 0.0) f++;
        first[i] = f;
    }
    for (int j = 0; j < n; j++) {
        double d = S[j * (size_t)n + j];
        for (int k = first[j]; k < j; k++) d -= S[j * (size_t)n + k] * S[j * (size_t)n + k];
        if (!(d > 0.0)) {
            free(first);
            return 0;
        }
        d = sqrt(d);
        S[j * (size_t)n + j] = d;
        for (int i = j + 1; i < n; i++) {
            if (first[i] > j) continue; /* S[i][j] is a structural zero and stays one */
            double v = S[i * (size_t)n + j];
            const int k0 = first[i] > first[j] ? first[i] : first[j];
            for (int k = k0; k < j; k++) v -= S[i * (size_t)n + k] * S[j * (size_t)n + k];
            S[i * (size_t)n + j] = v / d;
        }
    }
    for (int i = 0; i < n; i++) {
        double v = rhs[i];
        for (int k = first[i]; k < i; k++) v -= S[i * (size_t)n + k] * rhs[k];
        rhs[i] = v / S[i * (size_t)n + i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double v = rhs[i];
        for (int k = i + 1; k < n; k++) {
            if (first[k] > i) continue;
            v -= S[k * (size_t)n + i] * rhs[k];
        }
        rhs[i] = v / S[i * (size_t)n + i];
    }
    free(first);
    return 1;
}

static void inv3(const double* m, double* o) { /* Eigen 3x3 inverse: cofactors / determinant */
    const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    const double invdet = 1.0 / (m[0] * c00 + m[1] * c01 + m[2] * c02);
    o[0] = c00 * invdet; o[1] = (m[2] * m[7] - m[1] * m[8]) * invdet; o[2] = (m[1] * m[5] - m[2] * m[4]) * invdet;
    o[3] = c01 * invdet; o[4] = (m[0] * m[8] - m[2] * m[6]) * invdet; o[5] = (m[2] * m[3] - m[0] * m[5]) * invdet;
    o[6] = c02 * invdet; o[7] = (m[1] * m[6] - m[0] * m[7]) * invdet; o[8] = (m[0] * m[4] - m[1] * m[3]) * invdet;
}

/* BlockSolver::solve with Schur complement (block_solver.hpp:354-486); lambda already on the diagonals */
static int solve_schur(ba* s, const int* pt_first, const int* pt_next) {
    const orc_ba_problem* p = s->p;
    const int np = s->np, nl = s->nl, n = 6 * np;
    double* S = (double*)calloc((size_t)(n > 0 ? n : 1) * (size_t)(n > 0 ? n : 1), sizeof(double));
    double* coeff = (double*)calloc((size_t)(n + 3 * nl + 1), sizeof(double));
    double* Dinv = (double*)malloc(sizeof(double) * 9 * (size_t)(nl > 0 ? nl : 1));
    for (int i = 0; i < np; i++)
        for (int r = 0; r < 6; r++)
            for (int c = 0; c < 6; c++) S[(6 * i + r) * n + 6 * i + c] = s->Hpp[36 * (size_t)i + r * 6 + c];
    /* landmarks in hessian order; their edge lists (active edges of that point, insertion order) are chained */
    for (int il = 0; il < p->n_points; il++) {
        const int hl = s->pt_hidx[il];
        if (hl < 0) continue;
        double* Di = Dinv + 9 * (size_t)hl;
        inv3(s->Hll + 9 * (size_t)hl, Di);
        const double* bl = s->b + n + 3 * hl;
        double db[3];
        for (int r = 0; r < 3; r++) db[r] = Di[r * 3] * bl[0] + Di[r * 3 + 1] * bl[1] + Di[r * 3 + 2] * bl[2];
        for (int k1 = pt_first[il]; k1 >= 0; k1 = pt_next[k1]) {
            const int i1 = s->pose_hidx[p->edge_pose[s->active[k1]]];
            if (i1 < 0) continue;
            const double* Bi = s->Hpl + 18 * (size_t)k1;
            double BDinv[18];
            for (int r = 0; r < 6; r++)
                for (int c = 0; c < 3; c++)
                    BDinv[r * 3 + c] = Bi[r * 3] * Di[c] + Bi[r * 3 + 1] * Di[3 + c] + Bi[r * 3 + 2] * Di[6 + c];
            for (int r = 0; r < 6; r++) coeff[6 * i1 + r] += Bi[r * 3] * db[0] + Bi[r * 3 + 1] * db[1] + Bi[r * 3 + 2] * db[2];
            for (int k2 = pt_first[il]; k2 >= 0; k2 = pt_next[k2]) {
                const int i2 = s->pose_hidx[p->edge_pose[s->active[k2]]];
                if (i2 < i1) continue; /* upper blocks only (i2 >= i1), also skips fixed (-1) */
                const double* Bj = s->Hpl + 18 * (size_t)k2;
                for (int r = 0; r < 6; r++)
                    for (int c = 0; c < 6; c++)
                        S[(6 * i1 + r) * n + 6 * i2 + c] -=
                            BDinv[r * 3] * Bj[c * 3] + BDinv[r * 3 + 1] * Bj[c * 3 + 1] + BDinv[r * 3 + 2] * Bj[c * 3 + 2];
            }
        }
    }
    for (int r = 0; r < n; r++) /* the linear solver reads the upper triangle: mirror it */
        for (int c = r + 1; c < n; c++) S[c * n + r] = S[r * n + c];
    for (int i = 0; i < n; i++) s->x[i] = s->b[i] - coeff[i]; /* _bschur */
    const int ok = n == 0 ? 1 : cholesky_solve(S, s->x, n);
    if (ok) {
        /* cl = bl - Hpl^T xp ; xl = Dinv cl */
        double* cl = coeff + n;
        memcpy(cl, s->b + n, sizeof(double) * 3 * (size_t)nl);
        for (int k = 0; k < s->n_active; k++) {
            const int e = s->active[k];
            const int i1 = s->pose_hidx[p->edge_pose[e]];
            if (i1 < 0) continue;
            const int hl = s->pt_hidx[p->edge_point[e]];
            const double* W = s->Hpl + 18 * (size_t)k;
            for (int c = 0; c < 3; c++)
                for (int r = 0; r < 6; r++) cl[3 * hl + c] -= W[r * 3 + c] * s->x[6 * i1 + r];
        }
        for (int hl = 0; hl < nl; hl++) {
            const double* Di = Dinv + 9 * (size_t)hl;
            for (int r = 0; r < 3; r++)
                s->x[n + 3 * hl + r] = Di[r * 3] * cl[3 * hl] + Di[r * 3 + 1] * cl[3 * hl + 1] + Di[r * 3 + 2] * cl[3 * hl + 2];
        }
    }
    free(S); free(coeff); free(Dinv);
    return ok;
}

static void apply_update(ba* s) { /* SparseOptimizer::update -> oplus on every index-mapped vertex */
    const orc_ba_problem* p = s->p;
    for (int i = 0; i < p->n_poses; i++)
        if (s->pose_hidx[i] >= 0) orc_se3_exp_mul(s->x + 6 * s->pose_hidx[i], s->pose[i].q, s->pose[i].t);
    for (int i = 0; i < p->n_points; i++)
        if (s->pt_hidx[i] >= 0)
            for (int r = 0; r < 3; r++) s->pt[i][r] += s->x[6 * s->np + 3 * s->pt_hidx[i] + r];
}

/* SparseOptimizer::optimize + OptimizationAlgorithmLevenberg::solve
 * (sparse_optimizer.cpp:354-419; optimization_algorithm_levenberg.cpp:61-189) */
static int optimize(ba* s, int iterations) {
    const orc_ba_problem* p = s->p;
    if (s->np + s->nl == 0) return -1;
    int* pt_first = (int*)malloc(sizeof(int) * (size_t)(p->n_points > 0 ? p->n_points : 1));
    int* pt_last = (int*)malloc(sizeof(int) * (size_t)(p->n_points > 0 ? p->n_points : 1));
    int* pt_next = (int*)malloc(sizeof(int) * (size_t)(s->n_active > 0 ? s->n_active : 1));
    for (int i = 0; i < p->n_points; i++) pt_first[i] = pt_last[i] = -1;
    for (int k = 0; k < s->n_active; k++) {
        const int il = p->edge_point[s->active[k]];
        pt_next[k] = -1;
        if (pt_first[il] < 0) pt_first[il] = k; else pt_next[pt_last[il]] = k;
        pt_last[il] = k;
    }
    const int nx = 6 * s->np + 3 * s->nl;
    double* diag_p = (double*)malloc(sizeof(double) * 6 * (size_t)(s->np > 0 ? s->np : 1));
    double* diag_l = (double*)malloc(sizeof(double) * 3 * (size_t)(s->nl > 0 ? s->nl : 1));
    int done = 0, ok = 1;
    for (int it = 0; it < iterations && !terminate(s) && ok; it++) {
        compute_active_errors(s);
        double currentChi = active_robust_chi2(s), tempChi = currentChi;
        const double iniChi = currentChi;
        build_system(s);
        if (it == 0) { /* computeLambdaInit */
            double maxDiagonal = 0.;
            for (int i = 0; i < s->np; i++)
                for (int j = 0; j < 6; j++) maxDiagonal = fmax(fabs(s->Hpp[36 * (size_t)i + 7 * j]), maxDiagonal);
            for (int i = 0; i < s->nl; i++)
                for (int j = 0; j < 3; j++) maxDiagonal = fmax(fabs(s->Hll[9 * (size_t)i + 4 * j]), maxDiagonal);
            s->lambda = 1e-5 * maxDiagonal;
            s->ni = 2;
            s->nBad = 0;
        }
        double rho = 0;
        int qmax = 0;
        do {
            memcpy(s->pose_bak, s->pose, sizeof(se3) * (size_t)p->n_poses); /* push */
            memcpy(s->pt_bak, s->pt, sizeof(double) * 3 * (size_t)p->n_points);
            for (int i = 0; i < s->np; i++) /* setLambda(lambda, true) */
                for (int j = 0; j < 6; j++) {
                    diag_p[6 * i + j] = s->Hpp[36 * (size_t)i + 7 * j];
                    s->Hpp[36 * (size_t)i + 7 * j] += s->lambda;
                }
            for (int i = 0; i < s->nl; i++)
                for (int j = 0; j < 3; j++) {
                    diag_l[3 * i + j] = s->Hll[9 * (size_t)i + 4 * j];
                    s->Hll[9 * (size_t)i + 4 * j] += s->lambda;
                }
            const int ok2 = solve_schur(s, pt_first, pt_next);
            apply_update(s);
            for (int i = 0; i < s->np; i++) /* restoreDiagonal */
                for (int j = 0; j < 6; j++) s->Hpp[36 * (size_t)i + 7 * j] = diag_p[6 * i + j];
            for (int i = 0; i < s->nl; i++)
                for (int j = 0; j < 3; j++) s->Hll[9 * (size_t)i + 4 * j] = diag_l[3 * i + j];
            compute_active_errors(s);
            tempChi = active_robust_chi2(s);
            if (!ok2) tempChi = DBL_MAX;
            rho = currentChi - tempChi;
            double scale = 0.; /* computeScale */
            for (int j = 0; j < nx; j++) scale += s->x[j] * (s->lambda * s->x[j] + s->b[j]);
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && isfinite(tempChi)) {
                double alpha = 1. - pow((2 * rho - 1), 3);
                alpha = fmin(alpha, 2. / 3.);
                const double scaleFactor = fmax(1. / 3., alpha);
                s->lambda *= scaleFactor;
                s->ni = 2;
                currentChi = tempChi;
            } else {
                s->lambda *= s->ni;
                s->ni *= 2;
                memcpy(s->pose, s->pose_bak, sizeof(se3) * (size_t)p->n_poses); /* pop */
                memcpy(s->pt, s->pt_bak, sizeof(double) * 3 * (size_t)p->n_points);
            }
            qmax++;
            s->trials++;
        } while (rho < 0 && qmax < 10 && !terminate(s));
        done++;
        if (qmax == 10 || rho == 0) { ok = 0; continue; } /* Terminate */
        if ((iniChi - currentChi) * 1e3 < iniChi) s->nBad++; else s->nBad = 0;
        if (s->nBad >= 3) ok = 0;
    }
    free(pt_first); free(pt_last); free(pt_next); free(diag_p); free(diag_l);
    return done;
}

static double point_depth(const ba* s, int e) { /* isDepthPositive: map() with the CURRENT estimates */
    const int ip = s->p->edge_pose[e];
    double pc[3];
    quat_rotate(s->pose[ip].q, s->pt[s->p->edge_point[e]], pc);
    return pc[2] + s->pose[ip].t[2];
}

int orc_bundle_adjust(const orc_ba_problem* p, const orc_ba_options* opt, const volatile uint8_t* stop,
                      float* Tcw_out, float* Xw_out, uint8_t* edge_outlier, double* edge_chi2_out, orc_ba_info* info) {
    ba s;
    memset(&s, 0, sizeof(s));
    s.p = p;
    s.stop = stop;
    const size_t nP = (size_t)(p->n_poses > 0 ? p->n_poses : 1), nL = (size_t)(p->n_points > 0 ? p->n_points : 1);
    const size_t nE = (size_t)(p->n_edges > 0 ? p->n_edges : 1);
    s.pose = (se3*)malloc(sizeof(se3) * nP); s.pose_bak = (se3*)malloc(sizeof(se3) * nP);
    s.pt = malloc(sizeof(double) * 3 * nL); s.pt_bak = malloc(sizeof(double) * 3 * nL);
    s.intr = malloc(sizeof(double) * 4 * nP);
    s.level = (int*)calloc(nE, sizeof(int));
    s.err = calloc(nE, sizeof(double) * 2);
    s.pose_hidx = (int*)malloc(sizeof(int) * nP); s.pt_hidx = (int*)malloc(sizeof(int) * nL);
    s.active = (int*)malloc(sizeof(int) * nE);
    s.Hpp = (double*)malloc(sizeof(double) * 36 * nP); s.Hll = (double*)malloc(sizeof(double) * 9 * nL);
    s.Hpl = (double*)malloc(sizeof(double) * 18 * nE);
    s.b = (double*)malloc(sizeof(double) * (6 * nP + 3 * nL)); s.x = (double*)calloc(6 * nP + 3 * nL, sizeof(double));
    for (int i = 0; i < p->n_poses; i++) {
        orc_se3_from_Tcw(p->Tcw + 12 * (size_t)i, s.pose[i].q, s.pose[i].t);
        for (int k = 0; k < 4; k++) s.intr[i][k] = (double)p->intr[4 * (size_t)i + k];
    }
    for (int i = 0; i < p->n_points; i++)
        for (int k = 0; k < 3; k++) s.pt[i][k] = (double)p->Xw[3 * (size_t)i + k];
    s.robust = opt->robust;
    s.delta = (double)opt->huber_delta;
    orc_ba_info inf;
    memset(&inf, 0, sizeof(inf));

    if (terminate(&s)) {
        inf.aborted = 1; /* Optimizer.cc:631-633: return before optimising, nothing is written back */
    } else {
        init_stage(&s);
        compute_active_errors(&s);
        inf.chi2_initial = active_robust_chi2(&s);
        const int r1 = optimize(&s, opt->its_stage1);
        inf.iterations_stage1 = r1 > 0 ? r1 : 0;
        inf.chi2_final = active_robust_chi2(&s);
        int do_more = opt->its_stage2 > 0;
        if (terminate(&s)) { do_more = 0; inf.aborted = 1; }
        if (do_more) {
            for (int e = 0; e < p->n_edges; e++) { /* Optimizer.cc:644-656 */
                if (edge_chi2(&s, e) > (double)opt->chi2_threshold || !(point_depth(&s, e) > 0.0)) s.level[e] = 1;
            }
            s.robust = 0; /* setRobustKernel(nullptr) on every edge */
            init_stage(&s);
            const int r2 = optimize(&s, opt->its_stage2);
            inf.iterations_stage2 = r2 > 0 ? r2 : 0;
            inf.chi2_final = active_robust_chi2(&s);
        }
    }
    inf.lambda_final = s.lambda;
    inf.lm_trials = s.trials;
    for (int e = 0; e < p->n_edges; e++) { /* Optimizer.cc:682-695 */
        const double c = edge_chi2(&s, e);
        const int out = inf.aborted && s.trials == 0 ? 0 : (c > (double)opt->chi2_threshold || !(point_depth(&s, e) > 0.0));
        if (edge_outlier) edge_outlier[e] = (uint8_t)out;
        if (edge_chi2_out) edge_chi2_out[e] = c;
        inf.n_outliers += out;
    }
    for (int i = 0; i < p->n_poses; i++) orc_se3_to_Tcw(s.pose[i].q, s.pose[i].t, Tcw_out + 12 * (size_t)i);
    for (int i = 0; i < p->n_points; i++)
        for (int k = 0; k < 3; k++) Xw_out[3 * (size_t)i + k] = (float)s.pt[i][k];
    if (info) *info = inf;
    free(s.pose); free(s.pose_bak); free(s.pt); free(s.pt_bak); free(s.intr); free(s.level); free(s.err);
    free(s.pose_hidx); free(s.pt_hidx); free(s.active); free(s.Hpp); free(s.Hll); free(s.Hpl); free(s.b); free(s.x);
    return 0;
}

/* ================= Optimizer::PoseOptimization (code/src/Optimizer.cc:239-434) ================= */
typedef struct {
    int n;
    const double* Xw; const double* obs; const double* w; double K[4];
    double (*err)[2];
    uint8_t* level; /* 1 = excluded from the optimisation (outlier of the previous round) */
    int robust; double delta;
    se3 pose;
} po_t;

/* EdgeSE3ProjectXYZOnlyPose::computeError + linearizeOplus (types_six_dof_expmap.h:153-157, .cpp:266-288) */
static void po_edge(const po_t* s, int e, const se3* T, double* err, double* J) {
    double pc[3];
    quat_rotate(T->q, s->Xw + 3 * e, pc);
    pc[0] += T->t[0]; pc[1] += T->t[1]; pc[2] += T->t[2];
    if (err) {
        err[0] = s->obs[2 * e] - (pc[0] / pc[2] * s->K[0] + s->K[2]);
        err[1] = s->obs[2 * e + 1] - (pc[1] / pc[2] * s->K[1] + s->K[3]);
    }
    if (J) {
        const double x = pc[0], y = pc[1], invz = 1.0 / pc[2], invz_2 = invz * invz, fx = s->K[0], fy = s->K[1];
        J[0] = x * y * invz_2 * fx;        J[1] = -(1 + (x * x * invz_2)) * fx; J[2] = y * invz * fx;
        J[3] = -invz * fx;                 J[4] = 0;                            J[5] = x * invz_2 * fx;
        J[6] = (1 + y * y * invz_2) * fy;  J[7] = -x * y * invz_2 * fy;         J[8] = -x * invz * fy;
        J[9] = 0;                          J[10] = -invz * fy;                  J[11] = y * invz_2 * fy;
    }
}

static double po_chi2_edge(const po_t* s, int e) {
    return s->err[e][0] * (s->w[e] * s->err[e][0]) + s->err[e][1] * (s->w[e] * s->err[e][1]);
}

static void po_errors(po_t* s) {
    for (int e = 0; e < s->n; e++)
        if (!s->level[e]) po_edge(s, e, &s->pose, s->err[e], 0);
}

static double po_robust_chi2(const po_t* s) {
    double chi = 0, rho[3];
    for (int e = 0; e < s->n; e++) {
        if (s->level[e]) continue;
        if (s->robust) {
            orc_huber(po_chi2_edge(s, e), s->delta, rho);
            chi += rho[0];
        } else chi += po_chi2_edge(s, e);
    }
    return chi;
}

static int po_optimize(po_t* s, int iterations, int* its_done, int* trials) {
    int n_active = 0;
    for (int e = 0; e < s->n; e++) n_active += !s->level[e];
    if (n_active == 0) return -1; /* "0 vertices to optimize" */
    double H[36], b[6], x[6], lambda = -1, ni = 2;
    int nBad = 0, ok = 1;
    for (int it = 0; it < iterations && ok; it++) {
        po_errors(s);
        double currentChi = po_robust_chi2(s), tempChi = currentChi;
        const double iniChi = currentChi;
        memset(H, 0, sizeof(H));
        memset(b, 0, sizeof(b));
        for (int e = 0; e < s->n; e++) { /* BaseUnaryEdge::constructQuadraticForm (base_unary_edge.hpp:43-72) */
            if (s->level[e]) continue;
            double J[12];
            po_edge(s, e, &s->pose, 0, J);
            double w = s->w[e], r1 = 1.0;
            if (s->robust) {
                double rho[3];
                orc_huber(po_chi2_edge(s, e), s->delta, rho);
                r1 = rho[1];
            }
            const double wo = r1 * w;
            for (int r = 0; r < 6; r++) {
                b[r] -= r1 * (J[r] * (w * s->err[e][0]) + J[6 + r] * (w * s->err[e][1]));
                for (int c = 0; c < 6; c++) H[r * 6 + c] += J[r] * wo * J[c] + J[6 + r] * wo * J[6 + c];
            }
        }
        if (it == 0) {
            double maxDiagonal = 0.;
            for (int j = 0; j < 6; j++) maxDiagonal = fmax(fabs(H[7 * j]), maxDiagonal);
            lambda = 1e-5 * maxDiagonal;
            ni = 2;
            nBad = 0;
        }
        double rho = 0;
        int qmax = 0;
        do {
            const se3 backup = s->pose;
            double A[36];
            memcpy(A, H, sizeof(A));
            for (int j = 0; j < 6; j++) A[7 * j] += lambda;
            memcpy(x, b, sizeof(b));
            const int ok2 = cholesky_solve(A, x, 6); /* LinearSolverDense (LDLT), same solution */
            orc_se3_exp_mul(x, s->pose.q, s->pose.t);
            po_errors(s);
            tempChi = po_robust_chi2(s);
            if (!ok2) tempChi = DBL_MAX;
            rho = currentChi - tempChi;
            double scale = 0.;
            for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + b[j]);
            scale += 1e-3;
            rho /= scale;
            if (getenv("ORC_POSE_TRACE")) fprintf(stderr, "cpu trial %d lambda %.6e temp %.9e rho %.6e cur %.9e\n", *trials, lambda, tempChi, rho, currentChi);
            if (rho > 0 && isfinite(tempChi)) {
                double alpha = 1. - pow((2 * rho - 1), 3);
                alpha = fmin(alpha, 2. / 3.);
                lambda *= fmax(1. / 3., alpha);
                ni = 2;
                currentChi = tempChi;
            } else {
                lambda *= ni;
                ni *= 2;
                s->pose = backup;
            }
            qmax++;
            (*trials)++;
        } while (rho < 0 && qmax < 10);
        (*its_done)++;
        if (qmax == 10 || rho == 0) { ok = 0; continue; }
        if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
        if (nBad >= 3) ok = 0;
    }
    return 0;
}

int orc_pose_optimization(const float* Tcw12, const float* intr, int32_t n, const float* Xw, const float* obs,
                          const float* inv_sigma2, float* Tcw_out12, uint8_t* outlier, int32_t* info) {
    if (info) info[0] = info[1] = 0;
    if (n < 3) return 0; /* Optimizer.cc:358-359 */
    po_t s;
    memset(&s, 0, sizeof(s));
    s.n = n;
    double* X = (double*)malloc(sizeof(double) * 3 * (size_t)n);
    double* O = (double*)malloc(sizeof(double) * 2 * (size_t)n);
    double* W = (double*)malloc(sizeof(double) * (size_t)n);
    for (int i = 0; i < 3 * n; i++) X[i] = (double)Xw[i];
    for (int i = 0; i < 2 * n; i++) O[i] = (double)obs[i];
    for (int i = 0; i < n; i++) W[i] = (double)inv_sigma2[i];
    s.Xw = X; s.obs = O; s.w = W;
    for (int k = 0; k < 4; k++) s.K[k] = (double)intr[k];
    s.err = calloc((size_t)n, sizeof(double) * 2);
    s.level = (uint8_t*)calloc((size_t)n, 1);
    s.robust = 1;
    s.delta = (double)(float)sqrt(5.991); /* const float deltaMono = sqrt(5.991) */
    se3 init;
    orc_se3_from_Tcw(Tcw12, init.q, init.t);
    memset(outlier, 0, (size_t)n);
    const float chi2Mono = 5.991f;
    int nBad = 0, its = 0, trials = 0;
    for (int round = 0; round < 4; round++) {
        s.pose = init; /* vSE3->setEstimate(Converter::toSE3Quat(pFrame->mTcw)) */
        po_optimize(&s, 10, &its, &trials);
        nBad = 0;
        for (int e = 0; e < n; e++) {
            if (outlier[e]) po_edge(&s, e, &s.pose, s.err[e], 0); /* :364-366 */
            const float chi2 = (float)po_chi2_edge(&s, e);
            if (chi2 > chi2Mono) {
                outlier[e] = 1;
                s.level[e] = 1;
                nBad++;
            } else {
                outlier[e] = 0;
                s.level[e] = 0;
            }
        }
        if (round == 2) s.robust = 0;
        if (n < 10) break; /* optimizer.edges().size() < 10 */
    }
    orc_se3_to_Tcw(s.pose.q, s.pose.t, Tcw_out12);
    if (info) { info[0] = its; info[1] = trials; }
    free(X); free(O); free(W); free(s.err); free(s.level);
    return n - nBad;
}
