// Per-joint rotation conversions of the reference's MANO parameter head, written once on a generic scalar so that the
// same source gives values (float) and exact derivatives (forward-mode dual numbers) on host and device:
//   rot6d -> rotation matrix   common/myhand/decoder_lijun_mano.py:118-125 (ParamRegressor.rot6d_to_rotmat)
//   rotation matrix -> quaternion -> axis-angle   common/myhand/utils/comm.py:176-200, 250-324, 203-247 (kornia-derived)
//   axis-angle -> rotation matrix   common/utils/manolayer.py:32-48 (rodrigues_batch, angle = |axis| + 1e-8)
// The backward kernels seed one dual direction per input component (6 resp. 3 passes of a few dozen flops) and contract
// the Jacobian with the incoming gradient -- no hand-derived adjoint of the branchy quaternion code to get wrong.
// tests/test_pose_head.py compiles this header for the host and checks values and vector-Jacobian products against
// torch autograd through the reference's own functions.
#pragma once
#include <math.h>

#ifndef RIH_HD
#if defined(__HIPCC__)
#define RIH_HD __host__ __device__
#else
#define RIH_HD
#endif
#endif

template <int N>
struct RihDual {
    float v;
    float d[N];
    RIH_HD RihDual() : v(0.f) { for (int i = 0; i < N; ++i) d[i] = 0.f; }
    RIH_HD RihDual(float x) : v(x) { for (int i = 0; i < N; ++i) d[i] = 0.f; }
};
template <int N> RIH_HD inline RihDual<N> operator+(const RihDual<N>& a, const RihDual<N>& b) {
    RihDual<N> r; r.v = a.v + b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i]; return r;
}
template <int N> RIH_HD inline RihDual<N> operator-(const RihDual<N>& a, const RihDual<N>& b) {
    RihDual<N> r; r.v = a.v - b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i]; return r;
}
template <int N> RIH_HD inline RihDual<N> operator-(const RihDual<N>& a) {
    RihDual<N> r; r.v = -a.v; for (int i = 0; i < N; ++i) r.d[i] = -a.d[i]; return r;
}
template <int N> RIH_HD inline RihDual<N> operator*(const RihDual<N>& a, const RihDual<N>& b) {
    RihDual<N> r; r.v = a.v * b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r;
}
template <int N> RIH_HD inline RihDual<N> operator/(const RihDual<N>& a, const RihDual<N>& b) {
    RihDual<N> r; const float ib = 1.f / b.v; r.v = a.v * ib;
    for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib;
    return r;
}
template <int N> RIH_HD inline RihDual<N> rih_sqrt(const RihDual<N>& a) {
    // derivative 0 at 0, like torch.norm's sub-gradient (a zero axis-angle vector must not produce NaN gradients)
    RihDual<N> r; r.v = sqrtf(a.v); const float k = r.v > 0.f ? 0.5f / r.v : 0.f;
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * k;
    return r;
}
template <int N> RIH_HD inline RihDual<N> rih_sin(const RihDual<N>& a) {
    RihDual<N> r; r.v = sinf(a.v); const float c = cosf(a.v); for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * c; return r;
}
template <int N> RIH_HD inline RihDual<N> rih_cos(const RihDual<N>& a) {
    RihDual<N> r; r.v = cosf(a.v); const float s = -sinf(a.v); for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * s; return r;
}
template <int N> RIH_HD inline RihDual<N> rih_atan2(const RihDual<N>& y, const RihDual<N>& x) {
    RihDual<N> r; r.v = atan2f(y.v, x.v); const float q = 1.f / (x.v * x.v + y.v * y.v);
    for (int i = 0; i < N; ++i) r.d[i] = (x.v * y.d[i] - y.v * x.d[i]) * q;
    return r;
}
RIH_HD inline float rih_sqrt(float a) { return sqrtf(a); }
RIH_HD inline float rih_sin(float a) { return sinf(a); }
RIH_HD inline float rih_cos(float a) { return cosf(a); }
RIH_HD inline float rih_atan2(float y, float x) { return atan2f(y, x); }
RIH_HD inline float rih_val(float a) { return a; }
template <int N> RIH_HD inline float rih_val(const RihDual<N>& a) { return a.v; }

// F.normalize(v, dim=1): v / max(|v|, 1e-12)
template <typename T> RIH_HD inline void rih_normalize3(const T v[3], T out[3]) {
    T n = rih_sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (!(rih_val(n) > 1e-12f)) n = T(1e-12f);
    for (int i = 0; i < 3; ++i) out[i] = v[i] / n;
}

// x[6] viewed as (3, 2): a1 = x[:, 0], a2 = x[:, 1]  ->  R = [b1 | b2 | b3] (row-major R[3*i + j]), axis-angle aa[3]
template <typename T> RIH_HD inline void rih_rot6d_to_rotmat_aa(const T x[6], T R[9], T aa[3]) {
    const T a1[3] = {x[0], x[2], x[4]}, a2[3] = {x[1], x[3], x[5]};
    T b1[3], b2[3], u[3];
    rih_normalize3(a1, b1);
    const T dot = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
    for (int i = 0; i < 3; ++i) u[i] = a2[i] - dot * b1[i];
    rih_normalize3(u, b2);
    const T b3[3] = {b1[1] * b2[2] - b1[2] * b2[1], b1[2] * b2[0] - b1[0] * b2[2], b1[0] * b2[1] - b1[1] * b2[0]};
    for (int i = 0; i < 3; ++i) { R[3 * i] = b1[i]; R[3 * i + 1] = b2[i]; R[3 * i + 2] = b3[i]; }
    // rotation_matrix_to_quaternion (comm.py:280-323) works on rmat_t = R^T: m(i,j) = rmat_t[i][j] = R[j][i]
#define RIH_M(i, j) R[3 * (j) + (i)]
    const T one(1.f);
    const bool d2 = rih_val(RIH_M(2, 2)) < 1e-6f;
    const bool d0_d1 = rih_val(RIH_M(0, 0)) > rih_val(RIH_M(1, 1));
    const bool d0_nd1 = rih_val(RIH_M(0, 0)) < -rih_val(RIH_M(1, 1));
    T q[4], t;
    if (d2 && d0_d1) {
        t = one + RIH_M(0, 0) - RIH_M(1, 1) - RIH_M(2, 2);
        q[0] = RIH_M(1, 2) - RIH_M(2, 1); q[1] = t; q[2] = RIH_M(0, 1) + RIH_M(1, 0); q[3] = RIH_M(2, 0) + RIH_M(0, 2);
    } else if (d2) {
        t = one - RIH_M(0, 0) + RIH_M(1, 1) - RIH_M(2, 2);
        q[0] = RIH_M(2, 0) - RIH_M(0, 2); q[1] = RIH_M(0, 1) + RIH_M(1, 0); q[2] = t; q[3] = RIH_M(1, 2) + RIH_M(2, 1);
    } else if (d0_nd1) {
        t = one - RIH_M(0, 0) - RIH_M(1, 1) + RIH_M(2, 2);
        q[0] = RIH_M(0, 1) - RIH_M(1, 0); q[1] = RIH_M(2, 0) + RIH_M(0, 2); q[2] = RIH_M(1, 2) + RIH_M(2, 1); q[3] = t;
    } else {
        t = one + RIH_M(0, 0) + RIH_M(1, 1) + RIH_M(2, 2);
        q[0] = t; q[1] = RIH_M(1, 2) - RIH_M(2, 1); q[2] = RIH_M(2, 0) - RIH_M(0, 2); q[3] = RIH_M(0, 1) - RIH_M(1, 0);
    }
#undef RIH_M
    const T h = T(0.5f) / rih_sqrt(t);
    for (int i = 0; i < 4; ++i) q[i] = q[i] * h;
    // quaternion_to_angle_axis (comm.py:227-247)
    const T s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    const T s = rih_sqrt(s2);
    const T two_theta = T(2.f) * (rih_val(q[0]) < 0.f ? rih_atan2(-s, -q[0]) : rih_atan2(s, q[0]));
    const T k = (rih_val(s2) > 0.f) ? two_theta / s : T(2.f);
    for (int i = 0; i < 3; ++i) {
        aa[i] = q[i + 1] * k;
        if (rih_val(aa[i]) != rih_val(aa[i])) aa[i] = T(0.f);      // aa[isnan(aa)] = 0  (comm.py:199)
    }
}

// rodrigues_batch (common/utils/manolayer.py:32-48): R = I + sin(a) L + (1 - cos(a)) L L, a = |axis| + 1e-8
template <typename T> RIH_HD inline void rih_rodrigues(const T ax[3], T R[9]) {
    const T angle = rih_sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]) + T(1e-8f);
    const T e[3] = {ax[0] / angle, ax[1] / angle, ax[2] / angle};
    const T sn = rih_sin(angle), oc = T(1.f) - rih_cos(angle);
    const T zero(0.f);
    const T L[9] = {zero, -e[2], e[1], e[2], zero, -e[0], -e[1], e[0], zero};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            T ll = L[3 * i] * L[j] + L[3 * i + 1] * L[3 + j] + L[3 * i + 2] * L[6 + j];
            R[3 * i + j] = T(i == j ? 1.f : 0.f) + sn * L[3 * i + j] + oc * ll;
        }
}

// vector-Jacobian products: din = J^T dout, one dual direction per input component
RIH_HD inline void rih_rot6d_vjp(const float x[6], const float dR[9], const float daa[3], float dx[6]) {
    RihDual<6> xd[6], R[9], aa[3];
    for (int i = 0; i < 6; ++i) { xd[i] = RihDual<6>(x[i]); xd[i].d[i] = 1.f; }
    rih_rot6d_to_rotmat_aa(xd, R, aa);
    for (int i = 0; i < 6; ++i) {
        float g = 0.f;
        if (dR != nullptr)
            for (int k = 0; k < 9; ++k) g += dR[k] * R[k].d[i];
        if (daa != nullptr)
            for (int k = 0; k < 3; ++k) g += daa[k] * aa[k].d[i];
        dx[i] = g;
    }
}
RIH_HD inline void rih_rodrigues_vjp(const float ax[3], const float dR[9], float dax[3]) {
    RihDual<3> a[3], R[9];
    for (int i = 0; i < 3; ++i) { a[i] = RihDual<3>(ax[i]); a[i].d[i] = 1.f; }
    rih_rodrigues(a, R);
    for (int i = 0; i < 3; ++i) {
        float g = 0.f;
        for (int k = 0; k < 9; ++k) g += dR[k] * R[k].d[i];
        dax[i] = g;
    }
}

// nn.Hardswish and 3 * tanh (decoder_lijun_mano.py:121, 254-255)
RIH_HD inline float rih_hardswish(float x) { return x * fminf(fmaxf(x + 3.f, 0.f), 6.f) * (1.f / 6.f); }
// torch's hardswish_backward: 0 for x <= -3, 1 for x >= 3, x/3 + 1/2 in between
RIH_HD inline float rih_hardswish_grad(float x) { return x <= -3.f ? 0.f : (x < 3.f ? x * (1.f / 3.f) + 0.5f : 1.f); }
