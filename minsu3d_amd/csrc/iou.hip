// Proposal x instance IoU and mask labels for gfx950.  Replaces get_iou/get_iou.cu:12-38 and
// cal_iou_and_masklabel/cal_iou_and_masklabel.cu:14-140.
//
// The reference gives every (proposal, instance) pair its own thread that rescans the whole
// proposal: O(P*I*np) loads.  Here one workgroup owns a proposal, reads its points ONCE
// (coalesced prop_idx, gathered int16 labels) into an LDS histogram over instances, then writes
// the IoU row.  Counts are integers, the quotient is the reference's double expression
// (float)inter / ((float)(np + ninst - inter) + 1e-5) rounded to f32 -> bit-identical.
#include "common.h"
#include "../../include/minsu3d_hip.h"

namespace {

constexpr int IOU_THREADS = 256;
constexpr int MAX_LDS_BINS = 12288;  // 48 KB of counters; more instances -> tiled passes

template <bool ON_PRED>
__global__ __launch_bounds__(IOU_THREADS) void iou_kernel(int I, int P, const int *__restrict__ prop_idx,
                                                          const int *__restrict__ prop_off,
                                                          const int16_t *__restrict__ inst_labels,
                                                          const int *__restrict__ inst_pointnum,
                                                          float *__restrict__ iou,
                                                          const float *__restrict__ sigmoid)
{
    extern __shared__ int hist[];  // [bins] + 1 total
    const int bins = min(I, MAX_LDS_BINS);
    int *total_p = hist + bins;
    for (int p = blockIdx.x; p < P; p += gridDim.x) {
        const int s = prop_off[p], e = prop_off[p + 1];
        for (int k0 = 0; k0 < I; k0 += bins) {
            const int kb = min(bins, I - k0);
            for (int k = threadIdx.x; k < kb; k += IOU_THREADS) hist[k] = 0;
            if (threadIdx.x == 0) *total_p = 0;
            __syncthreads();
            int my_total = 0;
            for (int i = s + (int)threadIdx.x; i < e; i += IOU_THREADS) {
                if (ON_PRED && !(sigmoid[i] > 0.5f)) continue;  // .cu:51,63
                my_total++;
                const int lab = (int)inst_labels[prop_idx[i]] - k0;
                if (lab >= 0 && lab < kb) atomicAdd(&hist[lab], 1);
            }
            if (ON_PRED) {
                my_total = wave_sum(my_total);
                if (lane_id() == 0 && my_total) atomicAdd(total_p, my_total);
            }
            __syncthreads();
            const int total = ON_PRED ? *total_p : (e - s);
            for (int k = threadIdx.x; k < kb; k += IOU_THREADS) {
                const int inter = hist[k];
                const double den = (double)(float)(total + inst_pointnum[k0 + k] - inter) + 1e-5;
                iou[(size_t)p * I + k0 + k] = (float)((double)(float)inter / den);
            }
            __syncthreads();
        }
    }
}

// argmax over instances with the serial semantics of .cu:84-92 (init 0 / index 0, strict >,
// ignored classes skipped -> the FIRST maximal instance wins), then per-point labels.
__global__ __launch_bounds__(IOU_THREADS) void mask_label_kernel(int I, int P, int ignored_label, float iou_thr,
                                                                 const int *__restrict__ prop_idx,
                                                                 const int *__restrict__ prop_off,
                                                                 const int16_t *__restrict__ inst_labels,
                                                                 const int16_t *__restrict__ inst_cls,
                                                                 const float *__restrict__ iou,
                                                                 uint8_t *__restrict__ mask_label,
                                                                 uint8_t *__restrict__ mask_label_mask)
{
    __shared__ float s_v[IOU_THREADS / 64];
    __shared__ int s_i[IOU_THREADS / 64];
    for (int p = blockIdx.x; p < P; p += gridDim.x) {
        float best = 0.f;
        int bi = 0x7fffffff;  // "none": resolves to index 0 like the reference's max_ind = 0
        for (int k = threadIdx.x; k < I; k += IOU_THREADS) {
            const float v = iou[(size_t)p * I + k];
            if (v > best && inst_cls[k] != ignored_label) {  // ascending k per thread keeps the first
                best = v;
                bi = k;
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const float ov = __shfl_xor(best, d, 64);
            const int oi = __shfl_xor(bi, d, 64);
            if (ov > best || (ov == best && oi < bi)) {
                best = ov;
                bi = oi;
            }
        }
        if (lane_id() == 0) {
            s_v[wave_id()] = best;
            s_i[wave_id()] = bi;
        }
        __syncthreads();
        best = s_v[0];
        bi = s_i[0];
#pragma unroll
        for (int w = 1; w < IOU_THREADS / 64; w++)
            if (s_v[w] > best || (s_v[w] == best && s_i[w] < bi)) {
                best = s_v[w];
                bi = s_i[w];
            }
        if (bi == 0x7fffffff) bi = 0;
        if (best >= iou_thr) {
            for (int i = prop_off[p] + (int)threadIdx.x; i < prop_off[p + 1]; i += IOU_THREADS) {
                if ((int)inst_labels[prop_idx[i]] == bi) mask_label[i] = 1;
                mask_label_mask[i] = 1;
            }
        }
        __syncthreads();
    }
}

template <bool ON_PRED>
int launch_iou(int I, int P, const int *pi, const int *po, const int16_t *il, const int *pn, float *iou,
               const float *sg, ms3d_stream_t stream)
{
    if (I <= 0 || P <= 0) return 0;
    const int bins = I < MAX_LDS_BINS ? I : MAX_LDS_BINS;
    const size_t lds = (size_t)(bins + 1) * sizeof(int);
    iou_kernel<ON_PRED><<<P < 8192 ? P : 8192, IOU_THREADS, lds, (hipStream_t)stream>>>(I, P, pi, po, il, pn, iou, sg);
    MS3D_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" {

int ms3d_get_iou(int I, int P, const int *pi, const int *po, const int16_t *il, const int *pn, float *iou,
                 ms3d_stream_t stream)
{
    return launch_iou<false>(I, P, pi, po, il, pn, iou, nullptr, stream);
}
int ms3d_get_mask_iou_on_cluster(int I, int P, const int *pi, const int *po, const int16_t *il, const int *pn,
                                 float *iou, ms3d_stream_t stream)
{
    return launch_iou<false>(I, P, pi, po, il, pn, iou, nullptr, stream);
}
int ms3d_get_mask_iou_on_pred(int I, int P, const int *pi, const int *po, const int16_t *il, const int *pn,
                              float *iou, const float *sg, ms3d_stream_t stream)
{
    return launch_iou<true>(I, P, pi, po, il, pn, iou, sg, stream);
}
int ms3d_get_mask_label(int I, int P, int ignored_label, float iou_thr, const int *pi, const int *po,
                        const int16_t *il, const int16_t *ic, const float *iou, uint8_t *ml, uint8_t *mlm,
                        ms3d_stream_t stream)
{
    if (P <= 0) return 0;
    mask_label_kernel<<<P < 8192 ? P : 8192, IOU_THREADS, 0, (hipStream_t)stream>>>(I, P, ignored_label, iou_thr,
                                                                                   pi, po, il, ic, iou, ml, mlm);
    MS3D_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
