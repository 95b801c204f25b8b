// pose_convert.h — Converter::toSE3Quat(Tcw) and Converter::toCvMat(SE3Quat) (code/src/Converter.cc:37-47, 49-74; g2o
// se3quat.h:58-60, 269-274; Eigen's Quaterniond(Matrix3d)) as ONE sequence of double operations shared by the host (ba.cpp:
// the poses that cross the C ABI) and the device (match_kernels.hip: track_link_kernel hands the first tracking stage's pose
// to the second without the host in between).  Same operations in the same order, correctly rounded +, -, *, /, sqrt on both
// sides, -ffp-contract=off: the float pose the device forms is the float pose the host would have formed, to the bit
// (tests/test_track_chain_gpu.py: the linked stages equal the separate ones).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace so {

__host__ __device__ inline void quat_from_R_hd(const double* R, double* q) {  // Eigen::Quaterniond(Matrix3d), published algorithm
    double t = R[0] + R[4] + R[8];
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

// float [R|t] rows -> unit quaternion (x y z w, w >= 0) + translation
__host__ __device__ inline void pose_from_T12_hd(const float* T, double* q, double* t3) {
    const double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    quat_from_R_hd(R, q);
    if (q[3] < 0) {  // SE3Quat::normalizeRotation
        q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3];
    }
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
    t3[0] = T[3];
    t3[1] = T[7];
    t3[2] = T[11];
}

// unit quaternion + translation -> float [R|t] rows (to_homogeneous_matrix cast to float)
__host__ __device__ inline void pose_to_T12_hd(const double* q, const double* t3, float* T) {
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    T[0] = (float)(1 - (tyy + tzz)); T[1] = (float)(txy - twz);       T[2] = (float)(txz + twy);        T[3] = (float)t3[0];
    T[4] = (float)(txy + twz);       T[5] = (float)(1 - (txx + tzz)); T[6] = (float)(tyz - twx);        T[7] = (float)t3[1];
    T[8] = (float)(txz - twy);       T[9] = (float)(tyz + twx);       T[10] = (float)(1 - (txx + tyy)); T[11] = (float)t3[2];
}

}  // namespace so
