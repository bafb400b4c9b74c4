// Elastic distortion of a point cloud on the device.
// Replaces scipy.ndimage.convolve x 18 + scipy RegularGridInterpolator x 3 per call of the reference's
// minsu3d/util/transform.py:65-84 (called twice per training scene, general_dataset.py:118-120): the three float32
// noise grids (drawn by the caller with the host RNG, so that a seed gives the same augmentation as the reference)
// are box-blurred in place -- two rounds of zero-padded (1/3,1/3,1/3) taps along x, y, z, float64 sums stored back as
// float32 after every pass, as scipy does for float32 input -- and sampled trilinearly at the points in float64.
// Products and sums follow the order of the host restatement without FMA contraction; host and device agree to one
// ulp of the float64 result (measured 2.8e-14 voxels), far below anything the floor() quantisation can see.
#include "common.h"
#include "../../include/minsu3d_hip.h"

namespace {

__global__ void box_blur_axis_kernel(const float *__restrict__ in, float *__restrict__ out, int bx, int by, int bz, int axis,
                                     long total)
{
#pragma clang fp contract(off)  // __dmul_rn/__dadd_rn are plain operators in HIP's headers: without this they fuse into FMAs
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;  // over 3 grids
    if (e >= total) return;
    const long cell = e % ((long)bx * by * bz);
    const int z = (int)(cell % bz), y = (int)((cell / bz) % by), x = (int)(cell / ((long)bz * by));
    const int n = axis == 0 ? bx : (axis == 1 ? by : bz);
    const int i = axis == 0 ? x : (axis == 1 ? y : z);
    const long stride = axis == 0 ? (long)by * bz : (axis == 1 ? bz : 1);
    const double w = (double)(1.0f / 3.0f);
    const double hi = i + 1 < n ? (double)in[e + stride] : 0.0;
    const double mid = (double)in[e];
    const double lo = i > 0 ? (double)in[e - stride] : 0.0;
    double acc = __dadd_rn(0.0, __dmul_rn(w, hi));
    acc = __dadd_rn(acc, __dmul_rn(w, mid));
    acc = __dadd_rn(acc, __dmul_rn(w, lo));
    out[e] = (float)acc;
}

__global__ void elastic_sample_kernel(const double *__restrict__ xyz, int N, const float *__restrict__ noise, int bx, int by,
                                      int bz, double gran, double mag, double *__restrict__ out)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int n[3] = {bx, by, bz};
    double f[3];
    long i0[3];
    bool inside = true;
    const double step = __dmul_rn(2.0, gran);
    for (int a = 0; a < 3; a++) {
        const double lo = __dmul_rn(-(double)(n[a] - 1), gran);
        const double t = __ddiv_rn(__dsub_rn(xyz[(size_t)i * 3 + a], lo), step);
        inside = inside && t >= 0.0 && t <= (double)(n[a] - 1);
        long c = (long)floor(t);
        c = c < 0 ? 0 : (c > n[a] - 2 ? n[a] - 2 : c);
        i0[a] = c;
        f[a] = __dsub_rn(t, (double)c);
    }
    const long cells = (long)bx * by * bz;
    for (int comp = 0; comp < 3; comp++) {
        const float *g = noise + comp * cells;
        double s = 0.0;
        for (int dx = 0; dx < 2; dx++)
            for (int dy = 0; dy < 2; dy++)
                for (int dz = 0; dz < 2; dz++) {
                    const double wx = dx ? f[0] : __dsub_rn(1.0, f[0]);
                    const double wy = dy ? f[1] : __dsub_rn(1.0, f[1]);
                    const double wz = dz ? f[2] : __dsub_rn(1.0, f[2]);
                    const double wgt = __dmul_rn(__dmul_rn(wx, wy), wz);
                    const double v = (double)g[((i0[0] + dx) * by + (i0[1] + dy)) * bz + (i0[2] + dz)];
                    s = __dadd_rn(s, __dmul_rn(wgt, v));
                }
        if (!inside) s = 0.0;
        out[(size_t)i * 3 + comp] = __dadd_rn(xyz[(size_t)i * 3 + comp], __dmul_rn(s, mag));
    }
}

}  // namespace

extern "C" {

int ms3d_elastic_distort(const double *xyz, int N, float *noise, float *noise_tmp, int bx, int by, int bz, double gran,
                         double mag, double *out, ms3d_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (bx < 2 || by < 2 || bz < 2) return MS3D_E_UNSUPPORTED;
    const long total = 3L * bx * by * bz;
    float *src = noise, *dst = noise_tmp;
    for (int round = 0; round < 2; round++)
        for (int axis = 0; axis < 3; axis++) {
            box_blur_axis_kernel<<<ms3d_divup(total, 256), 256, 0, stream>>>(src, dst, bx, by, bz, axis, total);
            MS3D_LAUNCH_CHECK();
            float *t = src; src = dst; dst = t;
        }
    // six passes: the result is back in `noise`
    if (N > 0) {
        elastic_sample_kernel<<<ms3d_divup(N, 256), 256, 0, stream>>>(xyz, N, src, bx, by, bz, gran, mag, out);
        MS3D_LAUNCH_CHECK();
    }
    return 0;
}

}  // extern "C"
