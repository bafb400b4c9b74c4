// Adam step for ALL parameters of a model in one launch (gfx950).
// torch's fused Adam walks its tensor lists in 11 multi_tensor_apply launches of ~26 us each for the 250 parameter
// tensors of the m=16 networks (most of them 16..224-float BatchNorm vectors): 0.29 ms for 217 MB of traffic.
// Here the host keeps a chunk table (tensor, first element) and the four pointer tables on the device; a workgroup
// owns one 4096-element chunk of one tensor.
// Arithmetic = torch.optim.Adam (torch/optim/adam.py, _fused_adam / fused_adam_utils.cuh), f32:
//   g' = g + weight_decay * p;  m = lerp(m, g', 1 - beta1);  v = beta2 * v + (1 - beta2) * g' * g'
//   p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps),   bc_i = 1 - beta_i^step
#include "common.h"
#include "../../include/minsu3d_hip.h"

namespace {
constexpr int ADAM_CHUNK = 4096;

__global__ __launch_bounds__(256) void adam_step_kernel(const int2 *__restrict__ chunks, float *const *__restrict__ p_ptrs,
                                                       const float *const *__restrict__ g_ptrs,
                                                       float *const *__restrict__ m_ptrs, float *const *__restrict__ v_ptrs,
                                                       const long *__restrict__ sizes,
                                                       const float2 *__restrict__ coef, float lr_over_bc1,
                                                       float sqrt_bc2, float beta1, float beta2, float eps,
                                                       float weight_decay)
{
    const int2 c = chunks[blockIdx.x];
    if (coef) {     // per-tensor step counters: (lr / bias_correction1, sqrt(bias_correction2)) of THIS tensor
        const float2 k = coef[c.x];
        lr_over_bc1 = k.x;
        sqrt_bc2 = k.y;
    }
    float *p = p_ptrs[c.x];
    const float *g = g_ptrs[c.x];
    float *m = m_ptrs[c.x];
    float *v = v_ptrs[c.x];
    const long n = sizes[c.x];
    const long base = (long)c.y * ADAM_CHUNK;
    const long end = min(n, base + ADAM_CHUNK);
    auto update = [&](float &pp, float gg, float &mm, float &vv) {
        if (weight_decay != 0.f) gg = fmaf(weight_decay, pp, gg);
        mm = mm + (1.f - beta1) * (gg - mm);
        vv = beta2 * vv + (1.f - beta2) * gg * gg;
        const float denom = sqrtf(vv) / sqrt_bc2 + eps;
        pp -= lr_over_bc1 * mm / denom;
    };
    const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
    if (vec) {
        for (long i = base + 4 * threadIdx.x; i < end; i += 4 * 256) {
            if (i + 4 <= end) {
                float4 pp = *reinterpret_cast<float4 *>(p + i), mm = *reinterpret_cast<float4 *>(m + i),
                       vv = *reinterpret_cast<float4 *>(v + i);
                const float4 gg = *reinterpret_cast<const float4 *>(g + i);
                update(pp.x, gg.x, mm.x, vv.x); update(pp.y, gg.y, mm.y, vv.y);
                update(pp.z, gg.z, mm.z, vv.z); update(pp.w, gg.w, mm.w, vv.w);
                *reinterpret_cast<float4 *>(p + i) = pp;
                *reinterpret_cast<float4 *>(m + i) = mm;
                *reinterpret_cast<float4 *>(v + i) = vv;
            } else {
                for (long j = i; j < end; j++) update(p[j], g[j], m[j], v[j]);
            }
        }
    } else {
        for (long i = base + threadIdx.x; i < end; i += 256) update(p[i], g[i], m[i], v[i]);
    }
}
}  // namespace

extern "C" {

int ms3d_adam_chunk_elems(void) { return ADAM_CHUNK; }

int ms3d_adam_step(const int *chunks, int n_chunks, void *const *p_ptrs, const void *const *g_ptrs, void *const *m_ptrs,
                   void *const *v_ptrs, const long *sizes, float lr, float beta1, float beta2, float eps,
                   float weight_decay, double bias_correction1, double bias_correction2, ms3d_stream_t stream)
{
    if (n_chunks <= 0) return 0;
    if (bias_correction1 <= 0.0 || bias_correction2 <= 0.0) return MS3D_E_UNSUPPORTED;
    const float lr_over_bc1 = (float)((double)lr / bias_correction1);
    const float sqrt_bc2 = (float)sqrt(bias_correction2);
    adam_step_kernel<<<n_chunks, 256, 0, (hipStream_t)stream>>>(
        reinterpret_cast<const int2 *>(chunks), reinterpret_cast<float *const *>(p_ptrs),
        reinterpret_cast<const float *const *>(g_ptrs), reinterpret_cast<float *const *>(m_ptrs),
        reinterpret_cast<float *const *>(v_ptrs), sizes, nullptr, lr_over_bc1, sqrt_bc2, beta1, beta2, eps, weight_decay);
    MS3D_LAUNCH_CHECK();
    return 0;
}

int ms3d_adam_step_multi(const int *chunks, int n_chunks, void *const *p_ptrs, const void *const *g_ptrs,
                         void *const *m_ptrs, void *const *v_ptrs, const long *sizes, const float *coef, float beta1,
                         float beta2, float eps, float weight_decay, ms3d_stream_t stream)
{
    if (n_chunks <= 0) return 0;
    if (!coef) return MS3D_E_UNSUPPORTED;
    adam_step_kernel<<<n_chunks, 256, 0, (hipStream_t)stream>>>(
        reinterpret_cast<const int2 *>(chunks), reinterpret_cast<float *const *>(p_ptrs),
        reinterpret_cast<const float *const *>(g_ptrs), reinterpret_cast<float *const *>(m_ptrs),
        reinterpret_cast<float *const *>(v_ptrs), sizes, reinterpret_cast<const float2 *>(coef), 0.f, 1.f, beta1, beta2,
        eps, weight_decay);
    MS3D_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
