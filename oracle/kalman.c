/*
 * kalman.c -- CPU ORACLE (test infrastructure only) for `oat posifilt kalman`:
 *   oat::KalmanFilter2D::filter / initializeFilter / initializeStaticMatracies
 *   (src/positionfilter/KalmanFilter2D.cpp:95-210, members KalmanFilter2D.h:53-82)
 * over cv::KalmanFilter(4, 2, 0, CV_64F) of OpenCV 3.1.0 (modules/video/src/kalman.cpp,
 * [OCV-mem]: restated from the published algorithm, OpenCV is not in this image).
 *
 * PARITY UNPINNED (see oat_oracle.h).  One deliberate deviation: cv::KalmanFilter::correct solves
 * the 2x2 system (H P' H^T + R) X = H P' with cv::solve(DECOMP_SVD); here it is solved in closed
 * form (adjugate / determinant).  Both are backward-stable on this symmetric positive 2x2 system,
 * so they agree to rounding (tests allow 1e-9 px against an independent numpy restatement), not
 * necessarily bit for bit.
 *
 * Behaviour restated (including the reference's quirks, which a drop-in must keep):
 *  - state [x x' y y'], all double; cv::KalmanFilter::init leaves statePre/statePost/errorCovPre/
 *    errorCovPost zero, A = I, Q = I, R = I, H = 0 until initializeStaticMatracies runs.
 *  - kf_predicted_state_ and kf_meas_ are `cv::Mat_<double>{n, 1, CV_64F}` = n x 1 matrices FILLED
 *    WITH 6.0 (the (rows, cols, value) constructor; CV_64F == 6): before the first valid
 *    measurement the filter reports position (6, 6), velocity (6, 6), all flagged invalid.
 *  - filter(): a valid measurement resets the miss counter and (re)initialises the filter if it was
 *    not tracking; `if (not_found_count_ >= not_found_count_threshold_) found_ = false` runs on
 *    EVERY sample, so with the default --timeout 0 (threshold 0) the filter never tracks.
 *  - while tracking, every sample does predict() then correct(kf_meas_) -- on samples without a
 *    valid measurement the STALE last measurement is used for the correction.
 *  - the reported state is the PREDICTED one (statePre; `kf_predicted_state_ = kf_.predict()` shares
 *    statePre's buffer, so after initializeFilter() rewrote statePre the reported state follows it).
 *  - initializeFilter() sets errorCovPre = 1000 I, which the next predict() overwrites
 *    (errorCovPre = A errorCovPost A^T + Q) -- errorCovPost is NOT reset on re-initialisation.
 */
#include "oat_oracle.h"

#include <stdlib.h>
#include <string.h>

struct oat_kalman {
    double dt, sig_accel, sig_noise;
    int threshold;                 /* not_found_count_threshold_ = (int)(timeout / dt) */
    int found, not_found_count;
    double meas[2];                /* kf_meas_ */
    double reported[4];            /* kf_predicted_state_ until it starts aliasing statePre */
    int aliased;
    double A[16], Q[16], R[4];     /* H is the fixed selector of rows 0 and 2 */
    double statePre[4], statePost[4], Ppre[16], Ppost[16];
};

void oat_kalman_default_params(oat_kalman_params *p)
{
    p->dt = 0.02;            /* KalmanFilter2D.h:56 */
    p->timeout = 0.0;        /* threshold 0, KalmanFilter2D.h:73 */
    p->sigma_accel = 5.0;    /* :59 */
    p->sigma_noise = 0.0;    /* :60 */
}

oat_kalman *oat_kalman_create(const oat_kalman_params *p)
{
    oat_kalman *k = (oat_kalman *)calloc(1, sizeof *k);
    k->dt = p->dt; k->sig_accel = p->sigma_accel; k->sig_noise = p->sigma_noise;
    k->threshold = (int)(p->timeout / p->dt);          /* KalmanFilter2D.cpp:74-76 */
    for (int i = 0; i < 4; ++i) k->reported[i] = 6.0;
    k->meas[0] = k->meas[1] = 6.0;
    /* cv::KalmanFilter::init: A, Q, R identity; everything else zero */
    for (int i = 0; i < 4; ++i) k->A[i * 5] = k->Q[i * 5] = 1.0;
    k->R[0] = k->R[3] = 1.0;
    return k;
}

void oat_kalman_destroy(oat_kalman *k) { free(k); }

/* KalmanFilter2D.cpp:166-210 */
static void static_matrices(oat_kalman *k)
{
    const double dt = k->dt, sa = k->sig_accel;
    memset(k->A, 0, sizeof k->A);
    for (int i = 0; i < 4; ++i) k->A[i * 5] = 1.0;
    k->A[0 * 4 + 1] = dt;
    k->A[2 * 4 + 3] = dt;
    memset(k->Q, 0, sizeof k->Q);
    k->Q[0 * 4 + 0] = sa * sa * (dt * dt * dt * dt) / 4.0;
    k->Q[0 * 4 + 1] = sa * sa * (dt * dt * dt) / 2.0;
    k->Q[1 * 4 + 0] = sa * sa * (dt * dt * dt) / 2.0;
    k->Q[1 * 4 + 1] = sa * sa * (dt * dt);
    k->Q[2 * 4 + 2] = sa * sa * (dt * dt * dt * dt) / 4.0;
    k->Q[2 * 4 + 3] = sa * sa * (dt * dt * dt) / 2.0;
    k->Q[3 * 4 + 2] = sa * sa * (dt * dt * dt) / 2.0;
    k->Q[3 * 4 + 3] = sa * sa * (dt * dt);
    k->R[0] = k->R[3] = k->sig_noise * k->sig_noise;
    k->R[1] = k->R[2] = 0.0;
}

/* KalmanFilter2D.cpp:143-164 */
static void initialize_filter(oat_kalman *k)
{
    static_matrices(k);
    memset(k->Ppre, 0, sizeof k->Ppre);
    for (int i = 0; i < 4; ++i) k->Ppre[i * 5] = 1000.0;
    k->statePre[0] = k->meas[0]; k->statePre[1] = 0.0; k->statePre[2] = k->meas[1]; k->statePre[3] = 0.0;
    memcpy(k->statePost, k->statePre, sizeof k->statePre);
}

/* d (n x m) = a (n x 4) * b, b used as (4 x m) or, transposed, (m x 4); sums run k = 0..3 in order
 * (cv::gemm's generic double path), then `+ c` once */
static void mul4(const double *a, int n, const double *b, int m, int b_transposed, const double *c, double *d)
{
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) {
            double s = 0.0;
            for (int q = 0; q < 4; ++q) s += a[i * 4 + q] * (b_transposed ? b[j * 4 + q] : b[q * m + j]);
            d[i * m + j] = c ? s + c[i * m + j] : s;
        }
}

static void predict(oat_kalman *k)
{
    double t1[16];
    mul4(k->A, 4, k->statePost, 1, 0, NULL, k->statePre);       /* statePre = A statePost */
    mul4(k->A, 4, k->Ppost, 4, 0, NULL, t1);                    /* temp1 = A P */
    mul4(t1, 4, k->A, 4, 1, k->Q, k->Ppre);                     /* P' = temp1 A^T + Q */
    memcpy(k->statePost, k->statePre, sizeof k->statePre);
    memcpy(k->Ppost, k->Ppre, sizeof k->Ppre);
}

static void correct(oat_kalman *k)
{
    static const double H[8] = {1, 0, 0, 0, 0, 0, 1, 0};
    double t2[8], t3[4], t4[8], t5[2], hx[2];
    mul4(H, 2, k->Ppre, 4, 0, NULL, t2);                        /* temp2 = H P'            (2x4) */
    mul4(t2, 2, H, 2, 1, k->R, t3);                             /* temp3 = temp2 H^T + R   (2x2) */
    const double det = t3[0] * t3[3] - t3[1] * t3[2];           /* temp4 = temp3^-1 temp2  (2x4) */
    for (int j = 0; j < 4; ++j) {
        t4[j] = (t3[3] * t2[j] - t3[1] * t2[4 + j]) / det;
        t4[4 + j] = (t3[0] * t2[4 + j] - t3[2] * t2[j]) / det;
    }
    mul4(H, 2, k->statePre, 1, 0, NULL, hx);
    t5[0] = k->meas[0] - hx[0];                                 /* temp5 = z - H x' */
    t5[1] = k->meas[1] - hx[1];
    for (int i = 0; i < 4; ++i) {                               /* gain = temp4^T (4x2) */
        const double g0 = t4[i], g1 = t4[4 + i];
        k->statePost[i] = k->statePre[i] + (g0 * t5[0] + g1 * t5[1]);
        for (int j = 0; j < 4; ++j) k->Ppost[i * 4 + j] = k->Ppre[i * 4 + j] - (g0 * t2[j] + g1 * t2[4 + j]);
    }
}

/* KalmanFilter2D.cpp:95-141 */
void oat_kalman_filter(oat_kalman *k, int position_valid, double x, double y, oat_kalman_out *out)
{
    if (position_valid) {
        k->meas[0] = x; k->meas[1] = y;
        k->not_found_count = 0;
        if (!k->found) initialize_filter(k);
        k->found = 1;
    } else {
        k->not_found_count++;
    }
    if (k->not_found_count >= k->threshold) k->found = 0;
    if (k->found) {
        predict(k);
        k->aliased = 1;                 /* kf_predicted_state_ now shares statePre's buffer */
        correct(k);
    }
    const double *r = k->aliased ? k->statePre : k->reported;
    out->x = r[0]; out->vx = r[1]; out->y = r[2]; out->vy = r[3];
    out->position_valid = k->found;
    out->velocity_valid = k->found;
}


/* ------------------------------------------------------------------------------------------------
 * posifilt homography (HomographyTransform2D.cpp:62-107) -- ORACLE, test infrastructure.
 * cv::perspectiveTransform on one CV_64FC2 point with a 3x3 double matrix, OpenCV 3.1.0 core/matmul.cpp
 * perspectiveTransform_<double>:  w = x*m[6] + y*m[7] + m[8];  if (fabs(w) > FLT_EPSILON) { w = 1./w;
 * x' = (x*m[0] + y*m[1] + m[2])*w;  y' = (x*m[3] + y*m[4] + m[5])*w; } else x' = y' = 0.
 * The velocity goes through the same matrix with its offsets m[2], m[5] set to 0 (:79-89).  (Headings too,
 * followed by cv::normalize -- no detector on this path produces one.)  The position's unit becomes WORLD. */
#include <float.h>
#include <math.h>
static void perspective_point(const double *m, double *px, double *py)
{
    const double x = *px, y = *py;
    double w = x * m[6] + y * m[7] + m[8];
    if (fabs(w) > FLT_EPSILON) {
        w = 1. / w;
        *px = (x * m[0] + y * m[1] + m[2]) * w;
        *py = (x * m[3] + y * m[4] + m[5]) * w;
    } else
        *px = *py = 0;
}

void oat_homography_filter(const double h[9], int position_valid, double *x, double *y, int velocity_valid,
                           double *vx, double *vy)
{
    if (position_valid) perspective_point(h, x, y);
    if (velocity_valid) {
        double v[9];
        for (int i = 0; i < 9; i++) v[i] = h[i];
        v[2] = 0.0;
        v[5] = 0.0;
        perspective_point(v, vx, vy);
    }
}
