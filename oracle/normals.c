/* CPU oracle helper -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * The arithmetic behind legacy open3d.geometry.PointCloud.estimate_normals (pointcloud_alignment.py:27-28,
 * test/GICP1.py:77,148, test/check84.py:182): Open3D is un-vendored and un-pinned in the reference; this restates its published
 * algorithm [recalled, Open3D 0.13+ cpp/open3d/utility/Eigen.cpp ComputeCovariance and geometry/EstimateNormals.cpp
 * FastEigen3x3 / ComputeEigenvector0 / ComputeEigenvector1, the closed form of geometrictools' RobustEigenSymmetric3x3]:
 *
 *   covariance  one pass of nine cumulants over the RAW neighbour coordinates (sum x, y, z, xx, xy, xz, yy, yz, zz) in
 *               neighbour order (nearest first), divided by the count, then cov = E[ab] - E[a]E[b];
 *   normal      FastEigen3x3(cov): scale by the largest coefficient, eigenvalues from the trigonometric solution of the
 *               characteristic cubic, eigenvector of the best-conditioned extreme eigenvalue from the largest cross product of
 *               two rows of A - lambda I, the middle one from the 2x2 problem in its orthogonal complement, the third as their
 *               cross product; the eigenvector of the smallest eigenvalue is returned WITH THE SIGN THE FORMULA PRODUCES
 *               (legacy EstimateNormals keeps it unless the cloud already carries normals).
 *
 * PINNED by the reference's own recorded runs (tests/golden/output84, output: 163 frames, 2.14 M normals in the upstream
 * directories; tools/cpu_pin_normals.py writes the per-frame record): with the fused multiply-adds below the restatement
 * reproduces the recorded normals SIGN INCLUDED to <= 1 unit in the last place, most of them bit for bit.  The recorded runs come
 * from a build that contracts a*b+c into fused multiply-adds (the capture rig is an aarch64 Jetson; GCC contracts by default
 * there): without the fusions the same formulas are 1e-13 .. 1e-11 away, with them 1e-16.  WHICH operations that build fused is
 * not knowable from the source; the pattern below is the one that maximises bit-equality with the recorded normals (every site
 * is a switch in `cfg`, tools/cpu_pin_normals.py --search re-derives it), and it is what GCC's mult-add pass produces for these
 * expression shapes: a*b - c*d -> fma(a, b, -(c*d)); hand-written a0*b0 + a1*b1 + a2*b2 -> fma(a2, b2, fma(a0, b0, a1*b1));
 * Eigen's vectorised 3-vector dot -> fma(a2, b2, a0*b0 + a1*b1).
 */
#include <math.h>
#include <stdint.h>

enum { S_CROSS0, S_DOT0, S_LEN1, S_CROSS1, S_AMUL, S_MDOT, S_COMB, S_ONE, S_NORM, S_P, S_COF, S_DET, S_EVAL, S_CROSS2, S_ACC, S_COV, S_NSITES };
/* the contraction pattern of the build that recorded the fixtures (see the header) */
static const int32_t k_default_cfg[S_NSITES] = {1, 1, 1, 1, 2, 2, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1};

static inline double mulsub(double a, double b, double c, double d, int m) { /* a*b - c*d */
    if (m == 0) return a * b - c * d;
    if (m == 1) return fma(a, b, -(c * d));
    return fma(-c, d, a * b);
}
static inline double sum3(double x0, double y0, double x1, double y1, double x2, double y2, int m) {
    switch (m) {
    case 0: return (x0 * y0 + x1 * y1) + x2 * y2;
    case 1: return fma(x2, y2, x0 * y0 + x1 * y1);
    case 2: return fma(x2, y2, fma(x0, y0, x1 * y1));
    case 3: return fma(x2, y2, fma(x1, y1, x0 * y0));
    default: return fma(x0, y0, fma(x1, y1, x2 * y2));
    }
}
static inline double sum2(double x0, double y0, double x1, double y1, int m) {
    if (m == 0) return x0 * y0 + x1 * y1;
    if (m == 1) return fma(x0, y0, x1 * y1);
    return fma(x1, y1, x0 * y0);
}
static inline double madd(double a, double b, double c, int m) { return m ? fma(a, b, c) : a * b + c; }
static inline void cross3(const double a[3], const double b[3], double o[3], int m) {
    o[0] = mulsub(a[1], b[2], a[2], b[1], m);
    o[1] = mulsub(a[2], b[0], a[0], b[2], m);
    o[2] = mulsub(a[0], b[1], a[1], b[0], m);
}

/* A: scaled symmetric matrix as a00 a01 a02 a11 a12 a22 */
static void eigenvector0(const double A[6], double ev, double o[3], const int32_t *cfg) {
    const double r0[3] = {A[0] - ev, A[1], A[2]}, r1[3] = {A[1], A[3] - ev, A[4]}, r2[3] = {A[2], A[4], A[5] - ev};
    double c01[3], c02[3], c12[3];
    cross3(r0, r1, c01, cfg[S_CROSS0]);
    cross3(r0, r2, c02, cfg[S_CROSS0]);
    cross3(r1, r2, c12, cfg[S_CROSS0]);
    const double d0 = sum3(c01[0], c01[0], c01[1], c01[1], c01[2], c01[2], cfg[S_DOT0]);
    const double d1 = sum3(c02[0], c02[0], c02[1], c02[1], c02[2], c02[2], cfg[S_DOT0]);
    const double d2 = sum3(c12[0], c12[0], c12[1], c12[1], c12[2], c12[2], cfg[S_DOT0]);
    double dmax = d0;
    int imax = 0;
    if (d1 > dmax) { dmax = d1; imax = 1; }
    if (d2 > dmax) imax = 2;
    const double *c = imax == 0 ? c01 : imax == 1 ? c02 : c12;
    const double s = sqrt(imax == 0 ? d0 : imax == 1 ? d1 : d2);
    o[0] = c[0] / s; o[1] = c[1] / s; o[2] = c[2] / s;
}

static void eigenvector1(const double A[6], const double e0[3], double ev1, double o[3], const int32_t *cfg) {
    double U[3], V[3];
    if (fabs(e0[0]) > fabs(e0[1])) {
        const double il = 1 / sqrt(sum2(e0[0], e0[0], e0[2], e0[2], cfg[S_LEN1]));
        U[0] = -e0[2] * il; U[1] = 0; U[2] = e0[0] * il;
    } else {
        const double il = 1 / sqrt(sum2(e0[1], e0[1], e0[2], e0[2], cfg[S_LEN1]));
        U[0] = 0; U[1] = e0[2] * il; U[2] = -e0[1] * il;
    }
    cross3(e0, U, V, cfg[S_CROSS1]);
    const int am = cfg[S_AMUL], dm = cfg[S_MDOT];
    const double AU[3] = {sum3(A[0], U[0], A[1], U[1], A[2], U[2], am), sum3(A[1], U[0], A[3], U[1], A[4], U[2], am),
                          sum3(A[2], U[0], A[4], U[1], A[5], U[2], am)};
    const double AV[3] = {sum3(A[0], V[0], A[1], V[1], A[2], V[2], am), sum3(A[1], V[0], A[3], V[1], A[4], V[2], am),
                          sum3(A[2], V[0], A[4], V[1], A[5], V[2], am)};
    double m00 = sum3(U[0], AU[0], U[1], AU[1], U[2], AU[2], dm) - ev1;
    double m01 = sum3(U[0], AV[0], U[1], AV[1], U[2], AV[2], dm);
    double m11 = sum3(V[0], AV[0], V[1], AV[1], V[2], AV[2], dm) - ev1;
    const double a00 = fabs(m00), a01 = fabs(m01), a11 = fabs(m11);
    double cu, cv; /* result = cu * U - cv * V */
    if (a00 >= a11) {
        if (fmax(a00, a01) > 0) {
            if (a00 >= a01) { m01 /= m00; m00 = 1 / sqrt(madd(m01, m01, 1.0, cfg[S_ONE])); m01 *= m00; }
            else { m00 /= m01; m01 = 1 / sqrt(madd(m00, m00, 1.0, cfg[S_ONE])); m00 *= m01; }
            cu = m01; cv = m00;
        } else { o[0] = U[0]; o[1] = U[1]; o[2] = U[2]; return; }
    } else {
        if (fmax(a11, a01) > 0) {
            if (a11 >= a01) { m01 /= m11; m11 = 1 / sqrt(madd(m01, m01, 1.0, cfg[S_ONE])); m01 *= m11; }
            else { m11 /= m01; m01 = 1 / sqrt(madd(m11, m11, 1.0, cfg[S_ONE])); m11 *= m01; }
            cu = m11; cv = m01;
        } else { o[0] = U[0]; o[1] = U[1]; o[2] = U[2]; return; }
    }
    for (int i = 0; i < 3; i++) o[i] = mulsub(cu, U[i], cv, V[i], cfg[S_COMB]);
}

static void fast_eigen3x3(const double cov[9], double o[3], const int32_t *cfg) {
    double mx = cov[0];
    for (int i = 1; i < 9; i++) mx = cov[i] > mx ? cov[i] : mx;
    if (mx == 0) { o[0] = o[1] = o[2] = 0; return; }
    const double A[6] = {cov[0] / mx, cov[1] / mx, cov[2] / mx, cov[4] / mx, cov[5] / mx, cov[8] / mx};
    const double norm = sum3(A[1], A[1], A[2], A[2], A[4], A[4], cfg[S_NORM]);
    if (!(norm > 0)) { /* diagonal: the axis of the strictly smallest diagonal entry, z otherwise (on the UNscaled matrix) */
        const double a00 = A[0] * mx, a11 = A[3] * mx, a22 = A[5] * mx;   /* the original scales A back in place (A *= max_coeff) */
        o[0] = o[1] = o[2] = 0;
        if (a00 < a11 && a00 < a22) o[0] = 1;
        else if (a11 < a00 && a11 < a22) o[1] = 1;
        else o[2] = 1;
        return;
    }
    const double q = (A[0] + A[3] + A[5]) / 3;
    const double b00 = A[0] - q, b11 = A[3] - q, b22 = A[5] - q;
    double p;
    switch (cfg[S_P]) {
    case 0: p = sqrt((b00 * b00 + b11 * b11 + b22 * b22 + norm * 2) / 6); break;
    case 1: p = sqrt(fma(norm, 2.0, fma(b22, b22, fma(b00, b00, b11 * b11))) / 6); break;
    case 2: p = sqrt(fma(norm, 2.0, fma(b22, b22, fma(b11, b11, b00 * b00))) / 6); break;
    default: p = sqrt(fma(norm, 2.0, fma(b22, b22, b00 * b00 + b11 * b11)) / 6); break;
    }
    const double c00 = mulsub(b11, b22, A[4], A[4], cfg[S_COF]);
    const double c01 = mulsub(A[1], b22, A[4], A[2], cfg[S_COF]);
    const double c02 = mulsub(A[1], A[4], b11, A[2], cfg[S_COF]);
    double num;
    switch (cfg[S_DET]) {
    case 0: num = b00 * c00 - A[1] * c01 + A[2] * c02; break;
    case 1: num = fma(A[2], c02, fma(b00, c00, -(A[1] * c01))); break;
    case 2: num = fma(A[2], c02, fma(-A[1], c01, b00 * c00)); break;
    default: num = fma(A[2], c02, b00 * c00 - A[1] * c01); break;
    }
    const double det = num / (p * p * p);
    double half_det = det * 0.5;
    half_det = fmin(fmax(half_det, -1.0), 1.0);
    const double angle = acos(half_det) / 3.0;
    const double two_thirds_pi = 2.09439510239319549;
    const double beta2 = cos(angle) * 2;
    const double beta0 = cos(angle + two_thirds_pi) * 2;
    const double beta1 = -(beta0 + beta2);
    const double e0 = madd(p, beta0, q, cfg[S_EVAL]), e1 = madd(p, beta1, q, cfg[S_EVAL]), e2 = madd(p, beta2, q, cfg[S_EVAL]);
    double v0[3], v1[3], v2[3];
    if (half_det >= 0) {
        eigenvector0(A, e2, v2, cfg);
        if (e2 < e0 && e2 < e1) { o[0] = v2[0]; o[1] = v2[1]; o[2] = v2[2]; return; }
        eigenvector1(A, v2, e1, v1, cfg);
        if (e1 < e0 && e1 < e2) { o[0] = v1[0]; o[1] = v1[1]; o[2] = v1[2]; return; }
        cross3(v1, v2, o, cfg[S_CROSS2]);
    } else {
        eigenvector0(A, e0, v0, cfg);
        if (e0 < e1 && e0 < e2) { o[0] = v0[0]; o[1] = v0[1]; o[2] = v0[2]; return; }
        eigenvector1(A, v0, e1, v1, cfg);
        if (e1 < e0 && e1 < e2) { o[0] = v1[0]; o[1] = v1[1]; o[2] = v1[2]; return; }
        cross3(v0, v1, o, cfg[S_CROSS2]);
    }
}

/* pts [np,3]; idx [n,k] neighbour lists (nearest first), cnt[i] entries used; cov [n,9] out (identity when cnt < 3, as
 * EstimatePerPointCovariances).  cfg may be NULL (the pinned pattern). */
void r3d_oracle_cumulant_cov(const double *pts, const int64_t *idx, const int32_t *cnt, int64_t n, int32_t k, const int32_t *cfg, double *cov) {
    if (!cfg) cfg = k_default_cfg;
    const int fa = cfg[S_ACC], fc = cfg[S_COV];
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        double *C = cov + i * 9;
        if (cnt[i] < 3) {
            for (int j = 0; j < 9; j++) C[j] = (j % 4 == 0) ? 1.0 : 0.0;
            continue;
        }
        double c[9] = {0};
        for (int j = 0; j < cnt[i]; j++) {
            const double *p = pts + idx[i * k + j] * 3;
            c[0] += p[0]; c[1] += p[1]; c[2] += p[2];
            c[3] = madd(p[0], p[0], c[3], fa); c[4] = madd(p[0], p[1], c[4], fa); c[5] = madd(p[0], p[2], c[5], fa);
            c[6] = madd(p[1], p[1], c[6], fa); c[7] = madd(p[1], p[2], c[7], fa); c[8] = madd(p[2], p[2], c[8], fa);
        }
        for (int j = 0; j < 9; j++) c[j] /= (double)cnt[i];
#define R3D_COVSUB(e, a, b) (fc ? fma(-(a), (b), (e)) : (e) - (a) * (b))
        C[0] = R3D_COVSUB(c[3], c[0], c[0]);
        C[4] = R3D_COVSUB(c[6], c[1], c[1]);
        C[8] = R3D_COVSUB(c[8], c[2], c[2]);
        C[1] = C[3] = R3D_COVSUB(c[4], c[0], c[1]);
        C[2] = C[6] = R3D_COVSUB(c[5], c[0], c[2]);
        C[5] = C[7] = R3D_COVSUB(c[7], c[1], c[2]);
#undef R3D_COVSUB
    }
}

void r3d_oracle_fast_eigen3x3(const double *cov, int64_t n, const int32_t *cfg, double *normals) {
    if (!cfg) cfg = k_default_cfg;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) fast_eigen3x3(cov + i * 9, normals + i * 3, cfg);
}

int32_t r3d_oracle_normals_nsites(void) { return S_NSITES; }
void r3d_oracle_normals_default_cfg(int32_t *cfg) {
    for (int i = 0; i < S_NSITES; i++) cfg[i] = k_default_cfg[i];
}
