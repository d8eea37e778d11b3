/* examples/c_api_demo.c — libcalibba from plain C: refine the intrinsics of a pinhole + Brown-Conrady camera from four views
 * of a 3x3 target whose observations were rendered with a known camera (no distortion), then print the result.
 *   gcc -std=c99 -Iinclude examples/c_api_demo.c -Lcalibration_amd/lib -lcalibba -Wl,-rpath,$PWD/calibration_amd/lib -lm -o demo
 * The C ABI has no C++ / Eigen / torch types; this file is also compiled (not run) by the CPU test tier to keep the header C. */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "calibba.h"

int main(void) {
    enum { V = 4, N = 9 };
    const double K[10] = {1000, 1005, 640, 360, 0, 0, 0, 0, 0, 0};
    double X[V * N], Y[V * N], u[V * N], v[V * N], poses[V][7], intr[10];
    int64_t off[V + 1];
    int i, k;
    off[0] = 0;
    for (i = 0; i < V; ++i) {
        /* view pose: rotation about y by +-0.2 rad / about x by +-0.15 rad, 1 m in front of the camera */
        const double ay = (i % 2 ? 0.2 : -0.2), ax = (i / 2 ? 0.15 : -0.15);
        const double m[16] = {cos(ay), sin(ax) * sin(ay), -cos(ax) * sin(ay), 0, 0, cos(ax), sin(ax), 0,
                              sin(ay), -sin(ax) * cos(ay), cos(ax) * cos(ay), 0, 0.02 * i, -0.01 * i, 1.0, 1};  /* column-major 4x4 */
        cba_pose_from_matrix(m, poses[i]);
        for (k = 0; k < N; ++k) {
            const double x = 0.1 * (k % 3 - 1), y = 0.1 * (k / 3 - 1);
            const double px = m[0] * x + m[4] * y + m[12], py = m[1] * x + m[5] * y + m[13], pz = m[2] * x + m[6] * y + m[14];
            X[i * N + k] = x; Y[i * N + k] = y;
            u[i * N + k] = K[0] * px / pz + K[2];
            v[i * N + k] = K[1] * py / pz + K[3];
        }
        off[i + 1] = off[i] + N;
    }
    memcpy(intr, K, sizeof(K));
    intr[0] *= 0.97; intr[1] *= 1.03; intr[2] += 5; intr[3] -= 4;
    {
        cba_options o;
        cba_summary s;
        cba_status st;
        cba_options_default(&o);
        o.compute_covariance = 0;
        st = cba_optimize_intrinsics(CBA_CAMERA_PINHOLE_BC, V, off, X, Y, u, v, intr, &poses[0][0], &o, &s, NULL);
        if (st != CBA_OK) { fprintf(stderr, "libcalibba: %s\n", cba_last_error()); return 1; }
        printf("%s\nfx %.6f fy %.6f cx %.6f cy %.6f (success %d)\n", s.report, intr[0], intr[1], intr[2], intr[3], s.success);
    }
    return 0;
}
