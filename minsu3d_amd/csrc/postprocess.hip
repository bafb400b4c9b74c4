// Instance post-processing on the device.
// Replaces the dense [P, N] boolean-mask algebra of the reference's _get_pred_instances
// (minsu3d/model/pointgroup.py:197-265: masks.float() @ masks.float().T for the cross intersections, numpy argsort /
// delete loop for the greedy non-maximum suppression), which first copies every proposal to the host.
// Here proposals stay (cluster, point) pair lists:
//   cross intersection: pairs sorted by point; a point that belongs to m proposals contributes m*m integer atomics
//                       (m is 1-3 in practice) -> inter[P, P] with the proposal sizes on the diagonal;
//   greedy NMS:         one workgroup walks the proposals in descending score order, the threads of the group test
//                       the row of the picked proposal in parallel; IoU = inter / (n_a + n_b - inter) in float32,
//                       the same expression the reference evaluates on its float32 matrices.
#include "common.h"
#include "../../include/minsu3d_hip.h"

namespace {

__global__ void cross_intersection_kernel(const int *__restrict__ point, const int *__restrict__ cluster, int S, int P,
                                          int *__restrict__ inter)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= S) return;
    const int p = point[t], a = cluster[t];
    atomicAdd(&inter[(size_t)a * P + a], 1);
    for (int u = t + 1; u < S && point[u] == p; u++) {
        const int b = cluster[u];
        if (b == a) continue;  // a repeated (cluster, point) pair is one mask bit
        atomicAdd(&inter[(size_t)a * P + b], 1);
        atomicAdd(&inter[(size_t)b * P + a], 1);
    }
}

// duplicates of the same (cluster, point) pair must not inflate the diagonal: the caller passes unique pairs
// (ms3d_proposal_cross_intersection documents it); the kernel above only guards the off-diagonal terms.

__global__ __launch_bounds__(1024) void nms_greedy_kernel(const int *__restrict__ inter, const int *__restrict__ order, int P,
                                                          float threshold, unsigned char *__restrict__ suppressed,
                                                          int *__restrict__ pick, int *__restrict__ n_pick)
{
    __shared__ int s_n;
    if (threadIdx.x == 0) s_n = 0;
    for (int j = threadIdx.x; j < P; j += blockDim.x) suppressed[j] = 0;
    __syncthreads();
    for (int i = 0; i < P; i++) {
        const int a = order[i];
        if (suppressed[a]) continue;  // uniform: every thread reads the same byte after the barrier below
        const float na = (float)inter[(size_t)a * P + a];
        for (int j = threadIdx.x; j < P; j += blockDim.x) {
            if (j == a || suppressed[j]) continue;
            const float x = (float)inter[(size_t)a * P + j];
            const float nb = (float)inter[(size_t)j * P + j];
            if (x / (na + nb - x) > threshold) suppressed[j] = 1;
        }
        if (threadIdx.x == 0) pick[s_n++] = a;
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_pick = s_n;
}

}  // namespace

extern "C" {

int ms3d_proposal_cross_intersection(const int *pair_point, const int *pair_cluster, int S, int P, int *inter,
                                     ms3d_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (P <= 0) return 0;
    MS3D_CHECK(hipMemsetAsync(inter, 0, sizeof(int) * (size_t)P * P, stream));
    if (S <= 0) return 0;
    cross_intersection_kernel<<<ms3d_divup(S, 256), 256, 0, stream>>>(pair_point, pair_cluster, S, P, inter);
    MS3D_LAUNCH_CHECK();
    return 0;
}

int ms3d_nms_greedy(const int *inter, const int *order, int P, float threshold, unsigned char *suppressed_ws, int *pick,
                    int *n_pick, ms3d_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (P <= 0) {
        MS3D_CHECK(hipMemsetAsync(n_pick, 0, sizeof(int), stream));
        return 0;
    }
    nms_greedy_kernel<<<1, 1024, 0, stream>>>(inter, order, P, threshold, suppressed_ws, pick, n_pick);
    MS3D_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
