// Per-point losses of the backbone heads, forward AND the gradients, in three launches (gfx950).
// Replaces the ~30 torch operators of GeneralModel._loss / PTOffsetLoss (reference minsu3d/model/general_model.py:36-50,
// minsu3d/loss/pt_offset_loss.py:11-38) and their ~40 autograd operators: cross entropy with ignore_index -1 over the
// semantic scores, mean L1 norm of the offset error and mean negative cosine between predicted and ground-truth offset
// directions over the points that belong to an instance.  These run in the latency-bound stretch of a training step
// (behind the grouping, in front of the backward pass), where the GPU waits for every launch.
//
//   point_losses_kernel   one thread per point: log-softmax + pick (float, torch's formula: (x - max) - log sum exp(x - max)),
//                         offset terms, the UNNORMALISED gradients (softmax - onehot; sign(pred - gt); -d cos / d pred),
//                         per-block partial sums (double) of the three losses and the two valid counts
//   point_losses_finalize one block: partials in fixed order -> losses[3] and 1 / max(count, 1) [2]  (deterministic)
//   point_losses_scale    backward: gradients *= upstream scalar / count, both offset terms combined
#include "common.h"
#include "../../include/minsu3d_hip.h"

namespace {
constexpr int PL_THREADS = 256;

__device__ __forceinline__ double block_sum_256(double v, double *s_buf)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    __syncthreads();
    if (lane_id() == 0) s_buf[wave_id()] = v;
    __syncthreads();
    return s_buf[0] + s_buf[1] + s_buf[2] + s_buf[3];
}

// Score rows are C floats (80 bytes at C = 20): a thread that walked its own row in global memory would touch 64 different
// cache lines per load instruction (measured 185 us for 575k x 20).  A block stages its 256 rows through LDS instead --
// coalesced 16-byte loads in, the rows padded to an odd stride so that the per-thread row walk is conflict-free,
// the gradients written back into the same tile and streamed out coalesced.
__global__ __launch_bounds__(PL_THREADS) void point_losses_kernel(
    const float *__restrict__ scores, const short *__restrict__ labels, const float *__restrict__ pred,
    const float *__restrict__ centre, const float *__restrict__ xyz, const short *__restrict__ inst, long N, int C,
    float *__restrict__ d_scores, float *__restrict__ d_norm, float *__restrict__ d_dir, double *__restrict__ partial)
{
    extern __shared__ float s_tile[];            // [PL_THREADS][CP], CP = C | 1
    __shared__ double s_buf[4];
    const int CP = C | 1;
    double a_sem = 0.0, a_nsem = 0.0, a_norm = 0.0, a_dir = 0.0, a_noff = 0.0;
    const float eps = 1.1920928955078125e-07f;   // torch.finfo(float32).eps (pt_offset_loss.py:31-32)
    const long ntiles = (N + PL_THREADS - 1) / PL_THREADS;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long r0 = tile * PL_THREADS;
        const int rows = (int)((N - r0) < PL_THREADS ? (N - r0) : PL_THREADS);
        const long base = r0 * C;
        const int nel = rows * C;
        __syncthreads();
        for (int e = threadIdx.x; e < nel; e += PL_THREADS) s_tile[(e / C) * CP + (e % C)] = scores[base + e];
        __syncthreads();
        const long i = r0 + threadIdx.x;
        if (threadIdx.x < rows) {
            // ---- semantic cross entropy (ignore_index = -1)
            float *row = s_tile + threadIdx.x * CP;
            const int lab = labels[i];
            if (lab >= 0) {
                float m = -INFINITY;
                for (int c = 0; c < C; c++) m = fmaxf(m, row[c]);
                float se = 0.f;
                for (int c = 0; c < C; c++) se += expf(row[c] - m);
                const float lg = logf(se);
                a_sem -= (double)((row[lab] - m) - lg);
                a_nsem += 1.0;
                for (int c = 0; c < C; c++) row[c] = expf((row[c] - m) - lg) - (c == lab ? 1.f : 0.f);
            } else {
                for (int c = 0; c < C; c++) row[c] = 0.f;
            }
            // ---- offsets: points of an instance only
            float gn[3] = {0.f, 0.f, 0.f}, gd[3] = {0.f, 0.f, 0.f};
            if (inst[i] != -1) {
                float p[3], g[3];
#pragma unroll
                for (int k = 0; k < 3; k++) { p[k] = pred[3 * i + k]; g[k] = centre[3 * i + k] - xyz[3 * i + k]; }
                float dist = 0.f;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const float d = p[k] - g[k];
                    dist += fabsf(d);
                    gn[k] = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
                }
                const float ng = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
                const float np = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
                const float dg = fmaxf(ng, eps), dp = fmaxf(np, eps);
                const float gh[3] = {g[0] / dg, g[1] / dg, g[2] / dg};
                const float ph[3] = {p[0] / dp, p[1] / dp, p[2] / dp};
                const float cs = gh[0] * ph[0] + gh[1] * ph[1] + gh[2] * ph[2];
                a_norm += (double)dist;
                a_dir -= (double)cs;
                a_noff += 1.0;
                // d(-cos)/dp: p / max(|p|, eps) has Jacobian (I - u u^T) / |p| above eps (u = p / |p|), I / eps below it
#pragma unroll
                for (int k = 0; k < 3; k++)
                    gd[k] = np > eps ? -(gh[k] - cs * ph[k]) / np : -gh[k] / eps;
            }
#pragma unroll
            for (int k = 0; k < 3; k++) { d_norm[3 * i + k] = gn[k]; d_dir[3 * i + k] = gd[k]; }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < nel; e += PL_THREADS) d_scores[base + e] = s_tile[(e / C) * CP + (e % C)];
    }
    const double t0 = block_sum_256(a_sem, s_buf), t1 = block_sum_256(a_nsem, s_buf), t2 = block_sum_256(a_norm, s_buf),
                 t3 = block_sum_256(a_dir, s_buf), t4 = block_sum_256(a_noff, s_buf);
    if (threadIdx.x == 0) {
        double *o = partial + 5 * (size_t)blockIdx.x;
        o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3; o[4] = t4;
    }
}

__global__ __launch_bounds__(1024) void point_losses_finalize_kernel(const double *__restrict__ partial, int nblk,
                                                                    float *__restrict__ out /* [5] */)
{
    // thread t sums blocks t, t + 1024, ...; lanes fold with shuffles, the 16 wave sums in wave order: fixed order
    __shared__ double s_w[16][5];
    double s[5] = {0, 0, 0, 0, 0};
    for (int b = threadIdx.x; b < nblk; b += 1024)
#pragma unroll
        for (int t = 0; t < 5; t++) s[t] += partial[5 * (size_t)b + t];
#pragma unroll
    for (int t = 0; t < 5; t++)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) s[t] += __shfl_xor(s[t], d, 64);
    if (lane_id() == 0)
#pragma unroll
        for (int t = 0; t < 5; t++) s_w[wave_id()][t] = s[t];
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int t = 0; t < 5; t++) {
            double a = 0.0;
            for (int w = 0; w < 16; w++) a += s_w[w][t];
            s[t] = a;
        }
        const double nsem = s[1] > 1.0 ? s[1] : 1.0, noff = s[4] > 1.0 ? s[4] : 1.0;
        out[0] = (float)(s[0] / nsem);
        out[1] = (float)(s[2] / noff);
        out[2] = (float)(s[3] / noff);
        out[3] = (float)(1.0 / nsem);
        out[4] = (float)(1.0 / noff);
    }
}

__global__ __launch_bounds__(PL_THREADS) void point_losses_scale_kernel(float *__restrict__ d_scores, long n_scores,
                                                                       float *__restrict__ d_norm,
                                                                       const float *__restrict__ d_dir, long n_off,
                                                                       const float *__restrict__ out5,
                                                                       const float *__restrict__ g_sem,
                                                                       const float *__restrict__ g_norm,
                                                                       const float *__restrict__ g_dir)
{
    const float ks = (g_sem ? *g_sem : 0.f) * out5[3];
    const float kn = (g_norm ? *g_norm : 0.f) * out5[4], kd = (g_dir ? *g_dir : 0.f) * out5[4];
    const long stride = (long)gridDim.x * PL_THREADS, t0 = (long)blockIdx.x * PL_THREADS + threadIdx.x;
    if ((n_scores & 3) == 0 && ((uintptr_t)d_scores & 15) == 0) {
        float4 *d4 = reinterpret_cast<float4 *>(d_scores);
        for (long i = t0; i < (n_scores >> 2); i += stride) {
            float4 v = d4[i];
            v.x *= ks; v.y *= ks; v.z *= ks; v.w *= ks;
            d4[i] = v;
        }
    } else {
        for (long i = t0; i < n_scores; i += stride) d_scores[i] *= ks;
    }
    for (long i = t0; i < n_off; i += stride) d_norm[i] = kn * d_norm[i] + kd * d_dir[i];
}
}  // namespace

extern "C" {

int ms3d_point_losses_blocks(long N)
{
    long b = (N + PL_THREADS - 1) / PL_THREADS;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

int ms3d_point_losses_forward(const float *scores, const short *labels, const float *pred_offsets, const float *centre,
                              const float *xyz, const short *instance_ids, long N, int C, float *d_scores, float *d_norm,
                              float *d_dir, double *partial_ws, float *out5, ms3d_stream_t stream)
{
    const int nblk = ms3d_point_losses_blocks(N);
    if (C < 1 || C > 96) return MS3D_E_UNSUPPORTED;     // the 256-row tile lives in LDS
    point_losses_kernel<<<nblk, PL_THREADS, (size_t)PL_THREADS * (C | 1) * sizeof(float), (hipStream_t)stream>>>(
        scores, labels, pred_offsets, centre, xyz, instance_ids, N, C, d_scores, d_norm, d_dir, partial_ws);
    MS3D_LAUNCH_CHECK();
    point_losses_finalize_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(partial_ws, nblk, out5);
    MS3D_LAUNCH_CHECK();
    return 0;
}

int ms3d_point_losses_scale_grads(float *d_scores, long n_scores, float *d_norm, const float *d_dir, long n_off,
                                  const float *out5, const float *g_sem, const float *g_norm, const float *g_dir,
                                  ms3d_stream_t stream)
{
    long work = n_scores / 4 > n_off ? n_scores / 4 : n_off;
    long b = (work + PL_THREADS - 1) / PL_THREADS;
    const int nblk = (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
    point_losses_scale_kernel<<<nblk, PL_THREADS, 0, (hipStream_t)stream>>>(d_scores, n_scores, d_norm, d_dir, n_off, out5,
                                                                           g_sem, g_norm, g_dir);
    MS3D_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
