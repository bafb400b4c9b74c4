// Sparse 3-D convolution for gfx950 (CDNA4), the MinkowskiEngine arithmetic the reference backbone runs
// (call sites model/module/common.py:31,37,40,69,77, backbone.py:14; semantics SURVEY Appendix A.4/A.5/A.9).
//
// One kernel family serves k3 s1 (K = 27), k2 s2 (K = 8), transposed k2 s2 (K = 8, one-hot table) and
// k1 (K = 1, identity table), forward and backward-data, because every case is
//
//        out[i, :] = sum_k  act(in[nbr[k][i], :]) @ Weff[k]          (nbr = -1 -> no contribution)
//
// OUTPUT-STATIONARY: a 64-lane wave owns 16 output rows and all (<= 8 x 16) output columns, gathers the
// neighbour rows offset by offset and accumulates in registers; each output row is written exactly once,
// coalesced, without atomics (ME's gather-GEMM-scatter does one atomic scatter per pair).
//   * gather : lane (i = l & 15, q = l >> 4) reads 16 B = channels 16*ch + 4*q .. +3 of row nbr[k][row0+i];
//              the four q-lanes of a row cover one contiguous 64 B segment, the table is offset-major so the
//              16 indices of an offset are one 64 B read.
//   * math   : v_mfma_f32_16x16x4_f32 (exact f32, == an fmaf chain): the float4 just gathered feeds 4 MFMAs
//              (k-slot q <-> channel 4q+t), A never touches LDS.  Offsets none of the 16 rows has are skipped.
//   * weights: pre-permuted once per call into MFMA-fragment order Wf[k][ch][t][nb][lane] and kept in LDS
//              (up to 150 KB of the 160 KB/CU), so a B fragment is one conflict-free ds_read_b32 row.
//   * fusion : BatchNorm(+ReLU) of the INPUT is applied in the gather (a = max(0, x*scale+shift) on valid
//              rows only); residual add in the epilogue; backward-data applies the ReLU mask of the fused
//              BN and accumulates the two BN-backward channel sums in its epilogue.
// Backward-weight is a second kernel: dW[k] = sum_i act(in[nbr[k][i]])^T dout[i] on the same MFMA, rows as
// the reduction dimension, several offsets per wave sharing the dout fragment.
#include <stdlib.h>
#include <atomic>
#include <map>
#include <mutex>
#include <unordered_map>
#include "common.h"
#include "../../include/minsu3d_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int MAX_NBT = 8;             // column blocks (of 16) a wave accumulates
constexpr size_t LDS_BUDGET = 150 * 1024;
// dynamic-LDS ceiling of a kernel = the CU's 160 KB minus what the kernel declares statically
static hipError_t raise_lds_ceiling(const void *fn)
{
    hipFuncAttributes fa;
    hipError_t e = hipFuncGetAttributes(&fa, fn);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (int)fa.sharedSizeBytes);
}

int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

struct ConvArgs {
    const float *in;         // [Vin, Cin]
    const float *wf;         // fragment-major weights
    const float *wfs;        // the same weights in streamed order (prep_weights_kernel) or null
    const void *wfb;         // the three-piece bf16 image (write_bf3) or null
    const int *nbr;          // [K][Vout]
    float *out;              // [Vout, Cout]
    const float *pre_scale;  // [Cin] or null : a = max(0, x*scale + shift)
    const float *pre_shift;
    const float *residual;   // [Vout, Cout] or null, added in the epilogue
    const float *bias;       // [Cout] or null, added in the epilogue (dense per-point Linear layers: K = 1)
    // backward-data epilogue of a fused BN+ReLU (all null when unused)
    const float *bn_x;       // [Vout, Cout] input of the fused BN (forward)
    const float *bn_scale;   // [Cout]
    const float *bn_shift;   // [Cout]
    const float *bn_mean;    // [Cout]
    const float *bn_invstd;  // [Cout]
    float *bn_partial;       // [gridDim.x*gridDim.y][2][Cout] : sum(dz), sum(dz * xhat)   (bn_x != null)
                             //                               or sum(out), sum(out^2)       (out_stats != 0)
    int out_stats;           // forward: per-channel batch statistics of the OUTPUT ride in the epilogue
    int Vout, K, Cin, Cout;
    int NCH;    // ceil(Cin / 16)
    int NBtot;  // Cout / 16
    int G;      // offsets whose weights are LDS resident at once
    int GC;     // bf16x3 table walk: 32-channel chunks of those offsets resident at once (0 = all)
    int RT;     // small levels: 16-row tiles per block (0 / 1 = one)
    int ntiles;
    int pre_relu;
    const int *pl_tile_start;  // pair list of the table (ms3d_kmap_pairlist_build) or null
    const int *pl_entries;
    int dyn_picks;             // pair-list kernels: tiles picked off an LDS counter (MS3D_PL_DYNAMIC=1) instead of the fixed schedule
    int k1_path;               // K = 1: bit 0 = accumulate_k1 in the LDS-resident table walk, bit 1 = in the small-level kernel
    // weight-stationary kernels of the coarse levels (spconv_fwd_ws_kernel): partial-sum slabs [ws_ng][ntiles][NBtot][64] x 16 B,
    // one arrival counter per (row part, column slice) unit (zero between launches), offsets per group, row parts, tiles per part
    float *ws_slabs;
    unsigned *ws_cnt;
    int ws_ng, ws_kg, ws_R, ws_tpp;
};

// ------------------------------------------------------------------ weight permutation
// Wf[((k*NCH + ch)*4 + t)*NBtot + nb][lane] = Weff[k][c = 16ch + 4q + t][j = 16nb + (lane & 15)],  q = lane >> 4
// The STREAMED image holds the same numbers with the four steps t of a lane next to each other,
//     Wfs[((k*NCH + ch)*NBtot + nb)*64 + lane][t],
// so that a kernel that reads its weights straight from L2 fetches them 16 bytes per lane (spconv_fwd_pairstream_kernel).
__device__ __forceinline__ long stream_slot(long o, int NBtot)
{
    const int lane = (int)(o & 63);
    long r = o >> 6;
    const int nb = (int)(r % NBtot); r /= NBtot;
    const int t = (int)(r & 3); r >>= 2;   // r = k*NCH + ch
    return ((r * NBtot + nb) * 64 + lane) * 4 + t;
}

// ---- three-piece bf16 images (round 3) -----------------------------------------------------------------------
// f32 MFMA runs at 1/16 of the bf16 rate on gfx950 and there is no xf32.  A float splits EXACTLY into three bf16
// pieces (x = x0 + x1 + x2, 24 = 3 x 8 mantissa bits, each remainder computed in f32 without rounding), every
// bf16 x bf16 product is exact in f32, and the six products down to 2^-16 -- x0w0, x0w1, x1w0, x1w1, x0w2, x2w0 --
// carry a dot product to ~2^-23 relative: float32-grade results from v_mfma_f32_16x16x32_bf16 at 6 instructions per
// 32 input channels instead of 8 f32 instructions of twice the issue time (2.7x less matrix-pipe time).
// Image layout: Wb[(((k*NC32 + c32)*NB + nb)*3 + piece)*64 + lane][e]  (8 bf16 = 16 B per lane),
//               lane = (j & 15) + 16*g holds W[k][c = 32*c32 + 8*g + e][j = 16*nb + (lane & 15)], e = 0..7
// -- the B operand of the instruction for column block nb and 32-channel chunk c32, one ds_read_b128 per lane.
// It lives in the AUX slot behind the f32 image (where rectangular layers keep their streamed image).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__host__ __device__ inline bool bf3_dims_ok(int K, int Cin, int Cout)
{
    // input channels are padded to a multiple of 32 with zero weights (48 -> 64, 80 -> 96, 112 -> 128: the image still
    // fits the aux slot of 2n floats: 1.5 n * 64/48 = 2 n)
    return K > 1 && Cin % 16 == 0 && Cout % 16 == 0 && Cin >= 48 && Cout >= 48 && Cin <= 256 && Cout <= 256;
}

__device__ __forceinline__ void split3(float v, __bf16 &h0, __bf16 &h1, __bf16 &h2)
{
    h0 = (__bf16)v;
    const float r1 = v - (float)h0;   // exact
    h1 = (__bf16)r1;
    const float r2 = r1 - (float)h1;  // exact
    h2 = (__bf16)r2;
}

// Eight gathered floats of a lane -> the three bf16x8 MFMA operands, with the fused prologue: y = x * s + b (when
// `affine`), y = max(y, floor) (floor = 0 for ReLU, -inf otherwise: one instruction instead of max + select), absent
// neighbours / padded channels zeroed by `keep`.  The split works on PAIRS: v_cvt_pk_bf16_f32 rounds two floats at once
// and the rounded values come back as floats by a shift / a mask -- 11 instructions per pair against 16 of the
// element-wise split3 (same roundings, same pieces).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b)
{
    const bf16x2_t h = __builtin_convertvector((f32x2_t){a, b}, bf16x2_t);
    return *reinterpret_cast<const uint32_t *>(&h);
}
__device__ __forceinline__ void act_split8(const f32x4 &lo, const f32x4 &hi, bool affine, const f32x4 &s0, const f32x4 &s1,
                                           const f32x4 &b0, const f32x4 &b1, float floor_, int keep, bf16x8 &a0, bf16x8 &a1,
                                           bf16x8 &a2)
{
    float v[8];
#pragma unroll
    for (int t = 0; t < 4; t++) { v[t] = lo[t]; v[4 + t] = hi[t]; }
    if (affine) {
#pragma unroll
        for (int t = 0; t < 4; t++) {
            v[t] = fmaxf(fmaf(v[t], s0[t], b0[t]), floor_);
            v[4 + t] = fmaxf(fmaf(v[4 + t], s1[t], b1[t]), floor_);
        }
    }
    uint32_t q0[4], q1[4], q2[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const float x = __int_as_float(__float_as_int(v[2 * t]) & keep), y = __int_as_float(__float_as_int(v[2 * t + 1]) & keep);
        q0[t] = cvt_pk_bf16(x, y);
        const float rx = x - __uint_as_float(q0[t] << 16), ry = y - __uint_as_float(q0[t] & 0xffff0000u);     // exact
        q1[t] = cvt_pk_bf16(rx, ry);
        const float sx = rx - __uint_as_float(q1[t] << 16), sy = ry - __uint_as_float(q1[t] & 0xffff0000u);   // exact
        q2[t] = cvt_pk_bf16(sx, sy);
    }
    a0 = *reinterpret_cast<const bf16x8 *>(q0);
    a1 = *reinterpret_cast<const bf16x8 *>(q1);
    a2 = *reinterpret_cast<const bf16x8 *>(q2);
}

__device__ __forceinline__ void write_bf3(float *aux, int k, int c, int j, int Cin_e, int NB, float v)
{
    __bf16 *img = reinterpret_cast<__bf16 *>(aux);
    const int NC32 = (Cin_e + 31) >> 5, c32 = c >> 5, g = (c >> 3) & 3, e = c & 7, nb = j >> 4, jl = j & 15;
    const size_t base = ((((size_t)k * NC32 + c32) * NB + nb) * 3) * 64 + (jl + 16 * g);
    __bf16 h0, h1, h2;
    split3(v, h0, h1, h2);
    img[(base + 0) * 8 + e] = h0;
    img[(base + 64) * 8 + e] = h1;
    img[(base + 128) * 8 + e] = h2;
}

__global__ void prep_weights_kernel(const float *__restrict__ W, float *__restrict__ wf, int K, int Cin_e, int Cout_e,
                                    int NCH, int NBtot, int transpose, int mirror, float *__restrict__ wf2, int NCH2,
                                    int NBtot2, int mirror2, float *__restrict__ wfs, float *__restrict__ wfs2, int aux_kind)
{   // aux_kind: 1 = wfs / wfs2 take the streamed f32 image, 2 = the three-piece bf16 image
    const long total = (long)K * NCH * 4 * NBtot * 64;
    if (wf2) {
        // second image in the same launch: the backward-data operator (transposed, Cin/Cout swapped)
        const long total2 = (long)K * NCH2 * 4 * NBtot2 * 64;
        for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total2; o += (long)gridDim.x * blockDim.x) {
            const int lane = (int)(o & 63);
            long r = o >> 6;
            const int nb = (int)(r % NBtot2); r /= NBtot2;
            const int t = (int)(r & 3); r >>= 2;
            const int ch = (int)(r % NCH2);
            const int k = (int)(r / NCH2);
            const int c = 16 * ch + 4 * (lane >> 4) + t, j = 16 * nb + (lane & 15);  // c < Cout_e, j < Cin_e
            const int ks = mirror2 ? (K - 1 - k) : k;
            const float v = (c < Cout_e && j < Cin_e) ? W[((size_t)ks * Cin_e + j) * Cout_e + c] : 0.f;
            wf2[o] = v;
            if (wfs2 && aux_kind == 1) wfs2[stream_slot(o, NBtot2)] = v;
            if (wfs2 && aux_kind == 2 && c < Cout_e && j < Cin_e) {
                write_bf3(wfs2, k, c, j, Cout_e, NBtot2, v);
                if ((Cout_e & 16) && c >= Cout_e - 16) write_bf3(wfs2, k, c + 16, j, Cout_e, NBtot2, 0.f);   // zero pad
            }
        }
    }
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
        const int lane = (int)(o & 63);
        long r = o >> 6;
        const int nb = (int)(r % NBtot); r /= NBtot;
        const int t = (int)(r & 3); r >>= 2;
        const int ch = (int)(r % NCH);
        const int k = (int)(r / NCH);
        const int c = 16 * ch + 4 * (lane >> 4) + t, j = 16 * nb + (lane & 15);
        const int ks = mirror ? (K - 1 - k) : k;
        float v = 0.f;
        if (c < Cin_e && j < Cout_e)
            v = transpose ? W[((size_t)ks * Cout_e + j) * Cin_e + c]   // original layout [K][Cout_e(=Cin_o)][Cin_e(=Cout_o)]
                          : W[((size_t)ks * Cin_e + c) * Cout_e + j];
        wf[o] = v;
        if (wfs && aux_kind == 1) wfs[stream_slot(o, NBtot)] = v;
        if (wfs && aux_kind == 2 && c < Cin_e && j < Cout_e) {
            write_bf3(wfs, k, c, j, Cin_e, NBtot, v);
            if ((Cin_e & 16) && c >= Cin_e - 16) write_bf3(wfs, k, c + 16, j, Cin_e, NBtot, 0.f);              // zero pad
        }
    }
}

// Both images (forward, backward-data) of MANY layers in one launch: the per-layer launch is ~5 us of dispatch for a few
// KB of work, and a U-Net has ~90 of them per step.  descs[i].block_begin = first workgroup of layer i.
struct PrepDesc {
    const float *W;   // [K][Cin][Cout]
    float *wf, *wft;  // forward image, backward-data image (transposed, offsets mirrored if mirror_bwd);
                      // each is followed by its aux image: wf + n, wft + n  (n = ms3d_spconv_wf_floats; slot of 2n)
    int K, Cin, Cout, mirror_bwd, block_begin;
    int stream;       // aux images: 1 = streamed f32 (layers that can take spconv_fwd_pairstream_kernel), 2 = three-piece bf16
};
static_assert(sizeof(PrepDesc) == 48, "layout shared with the host-side descriptor table");

__global__ __launch_bounds__(256) void prep_weights_multi_kernel(const PrepDesc *__restrict__ descs, int n)
{
    int lo = 0, hi = n - 1;  // last layer with block_begin <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].block_begin <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const PrepDesc d = descs[lo];
    const int NCH = (d.Cin + 15) / 16, NB = (d.Cout + 15) / 16;
    const long total = (long)d.K * NCH * 4 * NB * 64;  // same element count for both images
    const long o = (long)(blockIdx.x - d.block_begin) * 256 + threadIdx.x;
    if (o >= total) return;
    const int lane = (int)(o & 63), q = lane >> 4, jl = lane & 15;
    {
        long r = o >> 6;
        const int nb = (int)(r % NB); r /= NB;
        const int t = (int)(r & 3); r >>= 2;
        const int ch = (int)(r % NCH), k = (int)(r / NCH);
        const int c = 16 * ch + 4 * q + t, j = 16 * nb + jl;
        const float v = (c < d.Cin && j < d.Cout) ? d.W[((size_t)k * d.Cin + c) * d.Cout + j] : 0.f;
        d.wf[o] = v;
        if (d.stream == 1) d.wf[total + stream_slot(o, NB)] = v;
    }
    {
        long r = o >> 6;
        const int nb = (int)(r % NCH); r /= NCH;  // the transposed operator: NCH2 = NB, NBtot2 = NCH
        const int t = (int)(r & 3); r >>= 2;
        const int ch = (int)(r % NB), k = (int)(r / NB);
        const int c = 16 * ch + 4 * q + t, j = 16 * nb + jl;  // c < Cout, j < Cin
        const int ks = d.mirror_bwd ? (d.K - 1 - k) : k;
        const float v = (c < d.Cout && j < d.Cin) ? d.W[((size_t)ks * d.Cin + j) * d.Cout + c] : 0.f;
        d.wft[o] = v;
        if (d.stream == 1) d.wft[total + stream_slot(o, NCH)] = v;
    }
    if (d.stream == 2) {
        // The three-piece bf16 images (layout: write_bf3), 16 bytes per piece and thread: thread (k, c32, nb, lane) reads the
        // 8 weights of its operand register (input channels 32 c32 + 8 (lane >> 4) + e, column 16 nb + (lane & 15)) and
        // stores three whole operands.  (One weight per thread scattered 2-byte stores over the image: 0.4 ms per HAIS
        // step.)  Channels past Cin are the zero padding of a half-filled 32-channel chunk.
        auto image = [&](float *aux, int Ci, int Co, int nb_tot, bool transposed) {
            const int NC32 = (Ci + 31) >> 5;
            if (o >= (long)d.K * NC32 * nb_tot * 64) return;
            const int g = lane >> 4;
            long r = o >> 6;
            const int nb = (int)(r % nb_tot); r /= nb_tot;
            const int c32 = (int)(r % NC32), k = (int)(r / NC32);
            const int ks = (transposed && d.mirror_bwd) ? (d.K - 1 - k) : k;
            const int j = 16 * nb + jl;
            bf16x8 p0, p1, p2;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int c = 32 * c32 + 8 * g + e;
                float v = 0.f;
                if (c < Ci && j < Co) v = transposed ? d.W[((size_t)ks * d.Cin + j) * d.Cout + c] : d.W[((size_t)ks * d.Cin + c) * d.Cout + j];
                __bf16 h0, h1, h2;
                split3(v, h0, h1, h2);
                p0[e] = h0; p1[e] = h1; p2[e] = h2;
            }
            bf16x8 *dst = reinterpret_cast<bf16x8 *>(aux) + ((((size_t)k * NC32 + c32) * nb_tot + nb) * 3) * 64 + lane;
            dst[0] = p0; dst[64] = p1; dst[128] = p2;
        };
        image(d.wf + total, d.Cin, d.Cout, NB, false);
        image(d.wft + total, d.Cout, d.Cin, NCH, true);     // the transposed operator: input side = Cout, columns = Cin
    }
}

// ------------------------------------------------------------------ forward / backward-data
constexpr int OG = 9;  // offsets processed together: 9 neighbour indices, then 9 row gathers in flight per wave (7 measured equal, 14 spills)

// table entries of offsets [g0, g0+OG) for one row; OR-mask instead of a select: a select lets the compiler sink
// the load into a branch + vmcnt(0)
template <int GSZ>
__device__ __forceinline__ void load_indices(const ConvArgs &p, int row, int g0, int k_hi, int (&idx)[GSZ])
{
    const bool row_ok = row >= 0 && row < p.Vout;
    const int safe_row = row_ok ? row : 0;
#pragma unroll
    for (int u = 0; u < GSZ; u++) {
        const int k = min(g0 + u, k_hi - 1);
        const int v = p.nbr[(size_t)k * p.Vout + safe_row];
        idx[u] = v | ((row_ok && g0 + u < k_hi) ? 0 : -1);
    }
}

// Offsets [k_lo, k_hi) of one 16-row tile.  `kw0` = index of offset k_lo inside the LDS weight image.
// (Prefetching the next group's indices across iterations was measured SLOWER -- 60 -> 70 us on the 16->16 level-0
// conv -- the extra live registers cost more than the saved round trip; each group loads its own indices.)
// Row loads are UNCONDITIONAL (index clamped to row 0, value zeroed afterwards): a branch around a gather makes the
// compiler drain vmcnt(0) before each one, which serialises the whole neighbourhood (measured: 27 x latency).
// GSZ: offsets per gather group -- OG where the registers allow it; the LDS table walk of 5+ column blocks (128 registers
// at 1024 threads) takes 7 / 6, which keeps its offset loop out of scratch
template <int NBT, bool ALIGNED, bool DIRECT = false, int GSZ = ((DIRECT || NBT < 5) ? OG : (NBT < 7 ? 7 : 6))>
__device__ __forceinline__ void accumulate_offsets(const ConvArgs &p, const float *__restrict__ sW, int k_lo, int k_hi,
                                                   int kw0, int my_row, int q, int nb0, f32x4 (&acc)[NBT])
{
    const int l = lane_id();
    int opaque0 = 0;
    asm volatile("" : "+s"(opaque0));
    for (int g0 = k_lo; g0 < k_hi; g0 += GSZ) {
        int idx[GSZ];
        load_indices(p, my_row, g0, k_hi, idx);
        bool any[GSZ];
#pragma unroll
        for (int u = 0; u < GSZ; u++) any[u] = __ballot(idx[u] >= 0) != 0ull;
        // rows (and the fused BN scale/shift) of one 16-channel chunk
        auto fetch = [&](int ch, f32x4 (&a)[GSZ], f32x4 &sc, f32x4 &sh) {
            const int c0 = 16 * ch + 4 * q;
#pragma unroll
            for (int u = 0; u < GSZ; u++) {
                const float *row = p.in + (size_t)max(idx[u], 0) * p.Cin + c0;
                if (ALIGNED) {
                    a[u] = *reinterpret_cast<const f32x4 *>(row);
                } else {
#pragma unroll
                    for (int t = 0; t < 4; t++) a[u][t] = (c0 + t < p.Cin) ? row[t] : 0.f;
                }
            }
            if (p.pre_scale) {
                if (ALIGNED) {
                    sc = *reinterpret_cast<const f32x4 *>(p.pre_scale + c0);
                    sh = *reinterpret_cast<const f32x4 *>(p.pre_shift + c0);
                } else {
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        sc[t] = (c0 + t < p.Cin) ? p.pre_scale[c0 + t] : 0.f;
                        sh[t] = (c0 + t < p.Cin) ? p.pre_shift[c0 + t] : 0.f;
                    }
                }
            }
        };
        // small levels (DIRECT): the next chunk's rows travel while this chunk is multiplied (one exposed round trip
        // per chunk otherwise, 4..14 chunks per launch); `opaque0` keeps the compiler from folding the prefetch back
        f32x4 a_next[GSZ], sc_next = {0.f, 0.f, 0.f, 0.f}, sh_next = sc_next;
        if (DIRECT) fetch(0, a_next, sc_next, sh_next);
        for (int ch = 0; ch < p.NCH; ch++) {
            f32x4 a[GSZ], s = {0.f, 0.f, 0.f, 0.f}, b = s;
            if (DIRECT) {
#pragma unroll
                for (int u = 0; u < GSZ; u++) a[u] = a_next[u];
                s = sc_next;
                b = sh_next;
                if (ch + 1 < p.NCH) fetch(ch + 1 + opaque0, a_next, sc_next, sh_next);
            } else {
                fetch(ch, a, s, b);
            }
            if (p.pre_scale) {
#pragma unroll
                for (int u = 0; u < GSZ; u++)
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        const float v = fmaf(a[u][t], s[t], b[t]);
                        a[u][t] = p.pre_relu ? fmaxf(v, 0.f) : v;
                    }
            }
#pragma unroll
            for (int u = 0; u < GSZ; u++) {
                const int keep = ~(idx[u] >> 31);  // absent neighbour (idx < 0) contributes nothing
#pragma unroll
                for (int t = 0; t < 4; t++) a[u][t] = __int_as_float(__float_as_int(a[u][t]) & keep);
            }
            if (DIRECT) {
                // small levels: B fragments straight from the global image (L2 resident), no LDS staging, and NO skip of
                // empty offsets: a uniform branch in front of the fragment loads makes the compiler issue them inside
                // the branch and wait on the spot (one L2 round trip per offset, 63 in a row on a 112-channel level).
                // The fragments of UB offsets are requested together (explicit register arrays: left alone the
                // scheduler keeps ~9 loads in flight, and a tiny level is exactly this chain of round trips).
                constexpr int UB0 = NBT <= 2 ? GSZ : (NBT <= 4 ? 3 : 1);
                constexpr int UB = UB0 < GSZ ? UB0 : GSZ;
#pragma unroll
                for (int u0 = 0; u0 < GSZ; u0 += UB) {
                    float wreg[UB][4][NBT];
#pragma unroll
                    for (int uu = 0; uu < UB; uu++) {
                        // offsets past K (last group) multiply zeros: clamp the read to the last real offset's image
                        const float *w = p.wf + ((size_t)((min(g0 + u0 + uu, p.K - 1) * p.NCH + ch) * 4) * p.NBtot + nb0) * 64 + l;
#pragma unroll
                        for (int t = 0; t < 4; t++)
#pragma unroll
                            for (int nb = 0; nb < NBT; nb++) wreg[uu][t][nb] = w[(size_t)(t * p.NBtot + nb) * 64];
                    }
                    __builtin_amdgcn_sched_barrier(0);  // keep the loads above the multiplies (the scheduler re-interleaves them)
#pragma unroll
                    for (int uu = 0; uu < UB; uu++)
#pragma unroll
                        for (int t = 0; t < 4; t++)
#pragma unroll
                            for (int nb = 0; nb < NBT; nb++)
                                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u0 + uu][t], wreg[uu][t][nb], acc[nb], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int u = 0; u < GSZ; u++) {
                    if (!any[u]) continue;
                    // LDS image holds only this block's NBT column blocks: [offset][ch][t][nb][lane]
                    const float *w = sW + (size_t)(((kw0 + (g0 - k_lo) + u) * p.NCH + ch) * 4) * NBT * 64 + l;
#pragma unroll
                    for (int t = 0; t < 4; t++) {
#pragma unroll
                        for (int nb = 0; nb < NBT; nb++)
                            acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][t], w[(t * NBT + nb) * 64], acc[nb], 0, 0, 0);
                    }
                }
            }
        }
    }
}

// K = 1 (the 1x1 projections on the skip paths, the per-point Linear layers): one table entry per row, so the gather
// parallelism the offset groups give the other layers has to come from the CHANNEL chunks -- four chunks of the row are
// requested together, then multiplied.  (Round 5.  Through accumulate_offsets a K = 1 layer issued eight clamped dummy
// gathers beside every real one, and the direct-B walk -- which multiplies every offset of its group, absent ones
// included -- did 9x the loads and MFMAs: 128 -> 256 at 12k rows took 140 us.)
template <int NBT, bool ALIGNED, bool DIRECT>
__device__ __forceinline__ void accumulate_k1(const ConvArgs &p, const float *__restrict__ sW, int my_row, int q, int nb0,
                                              f32x4 (&acc)[NBT])
{
    constexpr int CB = 4;   // chunks in flight
    const int l = lane_id();
    const bool row_ok = my_row >= 0 && my_row < p.Vout;
    const int idx = row_ok ? p.nbr[my_row] : -1;
    const int keep = ~(idx >> 31);
    if (__ballot(idx >= 0) == 0ull) return;
    const float *row = p.in + (size_t)max(idx, 0) * p.Cin + 4 * q;
    for (int ch0 = 0; ch0 < p.NCH; ch0 += CB) {
        f32x4 a[CB], sc[CB], sh[CB];
#pragma unroll
        for (int u = 0; u < CB; u++) {
            const int ch = min(ch0 + u, p.NCH - 1), c0 = 16 * ch + 4 * q;
            if (ALIGNED) {
                a[u] = *reinterpret_cast<const f32x4 *>(row + 16 * ch);
            } else {
#pragma unroll
                for (int t = 0; t < 4; t++) a[u][t] = (c0 + t < p.Cin) ? row[16 * ch + t] : 0.f;
            }
            sc[u] = sh[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (p.pre_scale) {
                if (ALIGNED) {
                    sc[u] = *reinterpret_cast<const f32x4 *>(p.pre_scale + c0);
                    sh[u] = *reinterpret_cast<const f32x4 *>(p.pre_shift + c0);
                } else {
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        sc[u][t] = (c0 + t < p.Cin) ? p.pre_scale[c0 + t] : 0.f;
                        sh[u][t] = (c0 + t < p.Cin) ? p.pre_shift[c0 + t] : 0.f;
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < CB; u++) {
            if (ch0 + u >= p.NCH) break;
            const int ch = ch0 + u;
            if (p.pre_scale) {
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const float v = fmaf(a[u][t], sc[u][t], sh[u][t]);
                    a[u][t] = p.pre_relu ? fmaxf(v, 0.f) : v;
                }
            }
#pragma unroll
            for (int t = 0; t < 4; t++) a[u][t] = __int_as_float(__float_as_int(a[u][t]) & keep);
            if (DIRECT) {
                const float *w = p.wf + ((size_t)(ch * 4) * p.NBtot + nb0) * 64 + l;
                float wreg[4][NBT];
#pragma unroll
                for (int t = 0; t < 4; t++)
#pragma unroll
                    for (int nb = 0; nb < NBT; nb++) wreg[t][nb] = w[(size_t)(t * p.NBtot + nb) * 64];
#pragma unroll
                for (int t = 0; t < 4; t++)
#pragma unroll
                    for (int nb = 0; nb < NBT; nb++)
                        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][t], wreg[t][nb], acc[nb], 0, 0, 0);
            } else {
                const float *w = sW + (size_t)(ch * 4) * NBT * 64 + l;   // LDS image of this block's slice: [ch][t][nb][lane]
#pragma unroll
                for (int t = 0; t < 4; t++)
#pragma unroll
                    for (int nb = 0; nb < NBT; nb++)
                        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][t], w[(t * NBT + nb) * 64], acc[nb], 0, 0, 0);
            }
        }
    }
}

// Statistics (round 5: bit-reproducible).  `slot` is the calling wave's OWN [2][NBT*16] area of the block's statistics
// (stat_slot): plain LDS read-modify-writes -- one writer per slot, a wave's LDS operations execute in order, a wave
// takes its tiles in a fixed order -- and stats_flush adds the slots up in wave order.  (Rounds 1-4 let all waves
// atomicAdd into one [2][Cout] area: the order of the float additions, and with it the last bits of every training-mode
// BatchNorm, changed from run to run.)
__host__ __device__ constexpr size_t stat_slot_floats(int nbt) { return 2 * (size_t)nbt * 16; }
constexpr int STAT_MAX_WAVES = 16;   // waves of the largest table-walk workgroup

template <int NBT>
__device__ __forceinline__ void stats_clear(float *s_part, int nslots)
{
    for (int t = threadIdx.x; t < nslots * (int)stat_slot_floats(NBT); t += blockDim.x) s_part[t] = 0.f;
}

// after the block's last store_tile: partial row of this block = the slots added in slot order; every blockIdx.y owns its
// own columns, the others stay zero and are summed away by the finalize kernel
template <int NBT>
__device__ __forceinline__ void stats_flush(const ConvArgs &p, const float *s_part, int nslots, int nb0)
{
    __syncthreads();
    float *dst = p.bn_partial + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 * p.Cout;
    for (int t = threadIdx.x; t < 2 * p.Cout; t += blockDim.x) {
        const int which = t >= p.Cout, c = t - which * p.Cout - 16 * nb0;  // column inside this block's slice
        float sum = 0.f;
        if (c >= 0 && c < NBT * 16)
            for (int w = 0; w < nslots; w++) sum += s_part[(size_t)w * stat_slot_floats(NBT) + which * NBT * 16 + c];
        dst[t] = sum;
    }
}

template <int NBT>
__device__ __forceinline__ void store_tile(const ConvArgs &p, int row0, int nb0, f32x4 (&acc)[NBT], float *slot)
{
    const int l = lane_id(), q = l >> 4, jl = l & 15;
#pragma unroll
    for (int nb = 0; nb < NBT; nb++) {
        const int j = 16 * (nb0 + nb) + jl;
        const bool j_ok = j < p.Cout;  // Cout need not be a multiple of 16 (padded column block)
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = row0 + 4 * q + r;  // C/D layout: row = 4*(lane>>4) + reg, col = lane & 15
            if (row < p.Vout && j_ok) {
                float v = acc[nb][r];
                const size_t o = (size_t)row * p.Cout + j;
                if (p.residual) v += p.residual[o];
                if (p.bias) v += p.bias[j];
                if (p.bn_x) {
                    // backward of the fused BN+ReLU that fed the forward conv: dz = da * [x*scale+shift > 0]
                    const float x = p.bn_x[o];
                    const float z = fmaf(x, p.bn_scale[j], p.bn_shift[j]);
                    v = (z > 0.f) ? v : 0.f;
                    s1 += v;
                    s2 += v * ((x - p.bn_mean[j]) * p.bn_invstd[j]);
                } else if (p.out_stats) {
                    s1 += v;
                    s2 = fmaf(v, v, s2);
                }
                p.out[o] = v;
            }
        }
        if (p.bn_x || p.out_stats) {
            // reduce over the 4 q-groups (same column), then into the wave's own slot (padded columns add zeros)
            s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
            if (q == 0) {
                slot[16 * nb + jl] += s1;
                slot[NBT * 16 + 16 * nb + jl] += s2;
            }
        }
    }
}

// RES: the whole image is LDS resident (persistent waves walk tiles) / streamed in groups of G offsets (one tile per wave).
// Two kernels, not one branch: with both paths in one body the epilogue's address arithmetic, hoisted out of the
// resident path's tile loop, went to scratch in BOTH (80-390 bytes per lane, profiles/r03_kernel_resources.txt).
template <int NBT, bool ALIGNED, bool RES>
__global__ __launch_bounds__(1024) void spconv_fwd_kernel(ConvArgs p)
{
    extern __shared__ float lds[];
    const int l = lane_id(), q = l >> 4;
    const int waves = blockDim.x >> 6;
    const int nb0 = blockIdx.y * NBT;
    float *sW = lds;
    float *s_part = lds + (size_t)p.G * p.NCH * 4 * NBT * 64;  // [waves] statistics slots when bn_x / out_stats
    const int slab4 = NBT * 16;                                    // float4s per (offset, ch, t) in the LDS image
    // copy offsets [k_lo, k_lo+cnt) of this block's column slice: contiguous NBT*64 floats per (k, ch, t)
    auto stage = [&](int k_lo, int cnt) {
        const int rows = cnt * p.NCH * 4;
        for (int e = threadIdx.x; e < rows * slab4; e += blockDim.x) {
            const int r = e / slab4, c4 = e - r * slab4;
            reinterpret_cast<f32x4 *>(sW)[e] =
                reinterpret_cast<const f32x4 *>(p.wf + ((size_t)k_lo * p.NCH * 4 + r) * p.NBtot * 64 + (size_t)nb0 * 64)[c4];
        }
    };

    const bool with_partial = p.bn_x != nullptr || p.out_stats != 0;
    if (with_partial) stats_clear<NBT>(s_part, waves);
    float *slot = s_part + (size_t)wave_id() * stat_slot_floats(NBT);
    // XCD-aware placement: blocks b, b+8, b+16.. share an XCD (dispatch is round-robin), give them adjacent tiles
    const int nblk = gridDim.x;
    const int per_xcd = (nblk + 7) / 8;
    int vb = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (nblk % 8 != 0) vb = blockIdx.x;  // only remap when it is a bijection

    if constexpr (RES) {
        // all weights resident: persistent waves walk a contiguous range of tiles
        stage(0, p.K);
        __syncthreads();
        const int total_waves = nblk * waves;
        const int chunk = (p.ntiles + total_waves - 1) / total_waves;
        const int wglobal = vb * waves + wave_id();
        const int t_begin = wglobal * chunk, t_end = min(p.ntiles, t_begin + chunk);
        for (int tile = t_begin; tile < t_end; tile++) {
            const int row0 = tile * 16;
            const int my_row = row0 + (l & 15);
            f32x4 acc[NBT];
#pragma unroll
            for (int nb = 0; nb < NBT; nb++) acc[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
            // K = 1 (the 1x1 projections and the per-point Linear layers): accumulate_k1
            if (p.K == 1 && (p.k1_path & 1)) accumulate_k1<NBT, ALIGNED, false>(p, sW, my_row, q, nb0, acc);
            else accumulate_offsets<NBT, ALIGNED>(p, sW, 0, p.K, 0, my_row, q, nb0, acc);
            // a per-tile view of the arguments: the epilogue's per-column address arithmetic is loop invariant, and hoisted
            // out of this loop it stayed live across the offset loop -- in scratch (up to 390 bytes per lane)
            ConvArgs pt = p;
            asm volatile("" : "+s"(pt.out), "+s"(pt.residual), "+s"(pt.bias), "+s"(pt.bn_x), "+s"(pt.bn_scale), "+s"(pt.bn_shift),
                         "+s"(pt.bn_mean), "+s"(pt.bn_invstd));
            store_tile<NBT>(pt, row0, nb0, acc, slot);
        }
    } else {
        // weights streamed through LDS in groups of G offsets; one tile per wave, accumulators stay in registers
        const int tile = vb * waves + wave_id();
        const int row0 = tile * 16;
        const int my_row = row0 + (l & 15);
        f32x4 acc[NBT];
#pragma unroll
        for (int nb = 0; nb < NBT; nb++) acc[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int g0 = 0; g0 < p.K; g0 += p.G) {
            const int gn = min(p.G, p.K - g0);
            __syncthreads();
            stage(g0, gn);
            __syncthreads();
            if (tile < p.ntiles) accumulate_offsets<NBT, ALIGNED>(p, sW, g0, g0 + gn, 0, my_row, q, nb0, acc);
        }
        if (tile < p.ntiles) store_tile<NBT>(p, row0, nb0, acc, slot);
    }
    if (with_partial) stats_flush<NBT>(p, s_part, waves, nb0);
}

// ---- table walk on the three-piece bf16 image (wide layers) -----------------------------------------------------
// Same grid and epilogue as spconv_fwd_kernel: a wave owns 16 output rows x NBT column blocks, the block's column slice
// of the weights goes through LDS.  Per offset and 32-channel chunk a lane gathers the 8 channels
// [32*c32 + 8*(lane >> 4), +8) of its row (two 16-byte loads), applies the fused BatchNorm / ReLU, splits the 8 floats
// into three bf16 pieces (exact) and issues six v_mfma_f32_16x16x32_bf16 per column block -- x0w0, x0w1, x1w0, x1w1,
// x0w2, x2w0 -- into the f32 accumulator: float32-grade (tests: 3e-6 against float64, as the f32 kernel).
// When the image does not fit, a STAGE is (OGB offsets) x (GC chunks): the gather rounds -- one per chunk, OGB rows in
// flight per lane -- stay full however wide the layer is.  (Staging all chunks of G offsets left 128 -> 128 with one
// offset per stage: four dependent one-row gather rounds between two barriers, 27 times.)  The first gather round of a
// stage is issued BEFORE the stage's barrier and weight copy, the index group of the next offsets one group ahead.
constexpr int OGB = 5;   // offsets per gather group: 10 x 16 B per lane in flight

// OG: offsets per gather round / stage (bf3_round_offsets(NBT): the most that stays inside the 128 registers of a 1024-thread block)
template <int NBT, int OG>
__global__ __launch_bounds__(1024) void spconv_fwd_bf3_kernel(ConvArgs p)
{
    extern __shared__ float lds[];
    const int l = lane_id(), g8 = l >> 4;
    const int waves = blockDim.x >> 6;
    const int nb0 = blockIdx.y * NBT;
    const int NC32 = (p.Cin + 31) >> 5;
    const int GC = p.GC > 0 ? p.GC : NC32;                                  // chunks per stage
    uint4 *sW = reinterpret_cast<uint4 *>(lds);                             // [offset in group][chunk in range][nb][piece][lane] x 16 B
    const int slab = NBT * 3 * 64;                                          // uint4 per (offset, c32) in the LDS image
    float *s_part = lds + (size_t)p.G * GC * slab * 4;                      // [waves] statistics slots when asked for
    const uint4 *img = reinterpret_cast<const uint4 *>(p.wfb);
    // offsets [k_lo, k_lo + cnt) x chunks [c_lo, c_lo + ccnt) of this block's column slice, copied by
    // global_load_lds_dwordx4: 64 lanes x 16 B straight into LDS at a wave-uniform base -- no registers, a handful of
    // address instructions per KB (the per-element index arithmetic of a register copy was a quarter of the kernel's VALU
    // work), completion on vmcnt (tools/probe/lds_direct_probe.hip checks the lane -> address mapping)
    auto stage = [&](int k_lo, int cnt, int c_lo, int ccnt) {
        uint4 *dst = sW;
        const int chunks = cnt * ccnt * NBT * 3;                            // 1 KB each
        for (int ch = wave_id(); ch < chunks; ch += waves) {
            const int r = ch / (NBT * 3), within = ch - r * (NBT * 3);
            const int kk = r / ccnt, cc = r - kk * ccnt;
            const uint4 *src = img + ((size_t)((k_lo + kk) * NC32 + c_lo + cc) * p.NBtot + nb0) * 3 * 64 + within * 64 + l;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(dst + (size_t)ch * 64), 16, 0, 0);
        }
    };
    const bool with_partial = p.bn_x != nullptr || p.out_stats != 0;
    if (with_partial) stats_clear<NBT>(s_part, waves);
    const int nblk = gridDim.x;
    const int per_xcd = (nblk + 7) / 8;
    int vb = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (nblk % 8 != 0) vb = blockIdx.x;  // only remap when it is a bijection
    const bool affine = p.pre_scale != nullptr;
    const float relu_floor = p.pre_relu ? 0.f : -INFINITY;

    // table entries of offsets [k0, k0 + OG) n [.., k_end) for this lane's row
    auto load_idx = [&](int k0, int k_end, int my_row, int (&idx)[OG]) {
        const bool row_ok = my_row < p.Vout;
        const int safe_row = row_ok ? my_row : 0;
#pragma unroll
        for (int u = 0; u < OG; u++) {
            const int k = min(k0 + u, p.K - 1);
            const int v = p.nbr[(size_t)k * p.Vout + safe_row];
            idx[u] = v | ((row_ok && k0 + u < k_end) ? 0 : -1);
        }
    };
    // one round: this lane's 8 channels of chunk c32 for the OG rows (unconditional, clamped: a branch around a gather
    // drains the counter)
    struct Round { f32x4 lo[OG], hi[OG]; };
    auto gather = [&](int c32, const int (&idx)[OG], Round &r) {
        const int cr = 32 * c32 + 8 * g8;
        const int c0 = cr < p.Cin ? cr : p.Cin - 8;
#pragma unroll
        for (int u = 0; u < OG; u++) {
            const float *row = p.in + (size_t)max(idx[u], 0) * p.Cin + c0;
            r.lo[u] = *reinterpret_cast<const f32x4 *>(row);
            r.hi[u] = *reinterpret_cast<const f32x4 *>(row + 4);
        }
    };
    // the MFMAs of one round against LDS slabs [(u * cstride + cslot)]
    auto compute = [&](int c32, int cslot, int cstride, const int (&idx)[OG], const bool (&any)[OG], const Round &r,
                       f32x4 (&acc)[NBT]) {
        const int cr = 32 * c32 + 8 * g8;
        const int c_keep = cr < p.Cin ? -1 : 0;     // Cin = 48, 80, 112: the upper half of the last chunk is padding
        const int c0 = cr < p.Cin ? cr : p.Cin - 8;
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, b0 = s0, b1 = s0;
        if (affine) {
            s0 = *reinterpret_cast<const f32x4 *>(p.pre_scale + c0); s1 = *reinterpret_cast<const f32x4 *>(p.pre_scale + c0 + 4);
            b0 = *reinterpret_cast<const f32x4 *>(p.pre_shift + c0); b1 = *reinterpret_cast<const f32x4 *>(p.pre_shift + c0 + 4);
        }
#pragma unroll
        for (int u = 0; u < OG; u++) {
            if (!any[u]) continue;
            bf16x8 a0, a1, a2;
            // absent neighbour (idx < 0) / padded channels contribute nothing
            act_split8(r.lo[u], r.hi[u], affine, s0, s1, b0, b1, relu_floor, ~(idx[u] >> 31) & c_keep, a0, a1, a2);
            const uint4 *w = sW + (size_t)(u * cstride + cslot) * slab + l;
#pragma unroll
            for (int nb = 0; nb < NBT; nb++) {
                const uint4 r0 = w[(nb * 3 + 0) * 64], r1 = w[(nb * 3 + 1) * 64], r2 = w[(nb * 3 + 2) * 64];
                const bf16x8 w0 = *reinterpret_cast<const bf16x8 *>(&r0), w1 = *reinterpret_cast<const bf16x8 *>(&r1),
                             w2 = *reinterpret_cast<const bf16x8 *>(&r2);
                // smallest terms first
                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, w0, acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, w2, acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, w1, acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, w0, acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, w1, acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, w0, acc[nb], 0, 0, 0);
            }
        }
    };

    {
        // one tile per wave, accumulators stay in registers; p.G = OG offsets x GC chunks per stage
        const int tile = vb * waves + wave_id();
        const int row0 = tile * 16, my_row = row0 + (l & 15);
        f32x4 acc[NBT];
#pragma unroll
        for (int nb = 0; nb < NBT; nb++) acc[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        int idx[OG];
        load_idx(0, min(p.G, p.K), my_row, idx);
        for (int k0 = 0; k0 < p.K; k0 += p.G) {
            const int gn = min(p.G, p.K - k0);
            int idx_n[OG];
            load_idx(k0 + p.G, min(k0 + 2 * p.G, p.K), my_row, idx_n);      // the next group's table entries, a whole group ahead
            bool any[OG];
#pragma unroll
            for (int u = 0; u < OG; u++) any[u] = __ballot(idx[u] >= 0) != 0ull;
            for (int c_lo = 0; c_lo < NC32; c_lo += GC) {
                const int ccnt = min(GC, NC32 - c_lo);
                Round r;
                gather(c_lo, idx, r);            // in flight across the barrier (and the weight copy)
                __syncthreads();
                stage(k0, gn, c_lo, ccnt);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share has landed (and its gathers)
                __syncthreads();
                if (tile < p.ntiles) compute(c_lo, 0, ccnt, idx, any, r, acc);
                for (int cc = 1; cc < ccnt; cc++) {
                    gather(c_lo + cc, idx, r);
                    if (tile < p.ntiles) compute(c_lo + cc, cc, ccnt, idx, any, r, acc);
                }
            }
#pragma unroll
            for (int u = 0; u < OG; u++) idx[u] = idx_n[u];
        }
        if (tile < p.ntiles) store_tile<NBT>(p, row0, nb0, acc, s_part + (size_t)wave_id() * stat_slot_floats(NBT));
    }
    if (with_partial) stats_flush<NBT>(p, s_part, waves, nb0);
}

// measured per column count (profiles/r03_fwd_experiments.txt section 9)
constexpr int bf3_round_offsets(int nbt) { return nbt == 2 || nbt == 4 || nbt == 6 ? 5 : 4; }

template <int NBT>
int launch_fwd_bf3(const ConvArgs &p, dim3 grid, int threads, size_t lds, hipStream_t stream)
{
    constexpr int OG = bf3_round_offsets(NBT);
    static const hipError_t attr = raise_lds_ceiling((const void *)spconv_fwd_bf3_kernel<NBT, OG>);
    MS3D_CHECK(attr);
    spconv_fwd_bf3_kernel<NBT, OG><<<grid, threads, lds, stream>>>(p);
    MS3D_LAUNCH_CHECK();
    return 0;
}

// Small levels (a few hundred to a few thousand rows at the bottom of the U-Net, 64..224 channels): the weight image
// (0.4 - 1.4 MB) dwarfs the activations, so staging it through LDS per block is the whole cost.  Here one block =
// one 16-row tile x NBT column blocks, its waves split the K offsets (9 each), read their B fragments directly from
// the L2-resident global image, and wave 0 sums the partial accumulators through LDS before the usual epilogue.
template <int NBT, bool ALIGNED>
__global__ __launch_bounds__(256) void spconv_fwd_small_kernel(ConvArgs p)
{
    extern __shared__ float lds[];
    const int l = lane_id(), q = l >> 4;
    const int waves = blockDim.x >> 6;  // = ceil(K / OG)
    const int nb0 = blockIdx.y * NBT;
    float *s_acc = lds;                                        // [(waves-1)][NBT][4][64]
    float *s_part = lds + (size_t)(waves - 1) * NBT * 256;     // one statistics slot (wave 0 finishes every tile) when bn_x / out_stats
    const bool with_partial = p.bn_x != nullptr || p.out_stats != 0;
    if (with_partial) stats_clear<NBT>(s_part, 1);
    // p.RT > 1: the geometry was laid out for the bf16x3 RT-tile kernel but this call has no bf16x3 image (the exact-f32
    // entry point on a wide layer): the same blocks take their RT tiles one after the other
    const int rt = p.RT > 1 ? p.RT : 1;
    for (int tt = 0; tt < rt; tt++) {
        const int tile = blockIdx.x * rt + tt;
        if (tile >= p.ntiles) break;
        const int row0 = tile * 16, my_row = row0 + (l & 15);
        f32x4 acc[NBT];
#pragma unroll
        for (int nb = 0; nb < NBT; nb++) acc[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int k_lo = wave_id() * OG, k_hi = min(p.K, k_lo + OG);
        // K = 1 (1x1 projections of the wide levels): accumulate_k1
        if (p.K == 1 && (p.k1_path & 2)) { if (k_lo < k_hi) accumulate_k1<NBT, ALIGNED, true>(p, nullptr, my_row, q, nb0, acc); }
        else if (k_lo < k_hi) accumulate_offsets<NBT, ALIGNED, true>(p, nullptr, k_lo, k_hi, 0, my_row, q, nb0, acc);
        if (tt > 0) __syncthreads();          // wave 0 is done with the previous tile's partials
        if (wave_id() > 0) {
#pragma unroll
            for (int nb = 0; nb < NBT; nb++)
#pragma unroll
                for (int r = 0; r < 4; r++) s_acc[((size_t)(wave_id() - 1) * NBT + nb) * 256 + r * 64 + l] = acc[nb][r];
        }
        __syncthreads();
        if (wave_id() == 0) {
            for (int w = 1; w < waves; w++)
#pragma unroll
                for (int nb = 0; nb < NBT; nb++)
#pragma unroll
                    for (int r = 0; r < 4; r++) acc[nb][r] += s_acc[((size_t)(w - 1) * NBT + nb) * 256 + r * 64 + l];
            store_tile<NBT>(p, row0, nb0, acc, s_part);
        }
    }
    if (with_partial) stats_flush<NBT>(p, s_part, 1, nb0);
}

// ---- small levels on the three-piece bf16 image: one block per (tile, column slice), its waves split the K offsets,
// B operands (16 bytes per lane and piece) straight from the L2-resident image, no LDS staging -- the bf16x3 counterpart
// of spconv_fwd_small_kernel.  The operands of the NEXT (offset, column group) are requested before the current group's
// MFMAs (two register sets, unconditional loads); an offset no row of the tile has skips its MFMAs only.
template <int NBT>
__global__ __launch_bounds__(256) void spconv_fwd_small_bf3_kernel(ConvArgs p)
{
    extern __shared__ float lds[];
    constexpr int NG = (NBT + 3) / 4;              // column groups of at most 4 blocks (12 operand loads in flight)
    constexpr int GB = NBT / NG, GR = NBT % NG;     // the first GR groups hold GB + 1 blocks
    const int l = lane_id(), g8 = l >> 4;
    const int waves = blockDim.x >> 6;  // = ceil(K / OG)
    const int nb0 = blockIdx.y * NBT;
    const int NC32 = (p.Cin + 31) >> 5;
    float *s_acc = lds;                                        // [(waves-1)][NBT][4][64]
    float *s_part = lds + (size_t)(waves - 1) * NBT * 256;     // [2*Cout] when bn_x / out_stats
    const bool with_partial = p.bn_x != nullptr || p.out_stats != 0;
    if (with_partial) stats_clear<NBT>(s_part, 1);      // wave 0 finishes the tile: one slot
    const int tile = blockIdx.x;
    const int row0 = tile * 16, my_row = row0 + (l & 15);
    const bool row_ok = my_row < p.Vout;
    const int safe_row = row_ok ? my_row : 0;
    f32x4 acc[NBT];
#pragma unroll
    for (int nb = 0; nb < NBT; nb++) acc[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const uint4 *img = reinterpret_cast<const uint4 *>(p.wfb);
    const int k_lo = wave_id() * OG, k_hi = min(p.K, k_lo + OG);
    for (int k0 = k_lo; k0 < k_hi; k0 += OGB) {
        int idx[OGB];
#pragma unroll
        for (int u = 0; u < OGB; u++) {
            const int k = min(k0 + u, k_hi - 1);
            const int v = p.nbr[(size_t)k * p.Vout + safe_row];
            idx[u] = v | ((row_ok && k0 + u < k_hi) ? 0 : -1);
        }
        bool any[OGB];
#pragma unroll
        for (int u = 0; u < OGB; u++) any[u] = __ballot(idx[u] >= 0) != 0ull;
        for (int c32 = 0; c32 < NC32; c32++) {
            const int cr = 32 * c32 + 8 * g8;
            const int c_keep = cr < p.Cin ? -1 : 0;
            const int c0 = cr < p.Cin ? cr : p.Cin - 8;
            f32x4 lo[OGB], hi[OGB];
#pragma unroll
            for (int u = 0; u < OGB; u++) {
                const float *row = p.in + (size_t)max(idx[u], 0) * p.Cin + c0;
                lo[u] = *reinterpret_cast<const f32x4 *>(row);
                hi[u] = *reinterpret_cast<const f32x4 *>(row + 4);
            }
            const float *ps = p.pre_scale ? p.pre_scale : p.in, *pb = p.pre_scale ? p.pre_shift : p.in;
            const f32x4 s0 = *reinterpret_cast<const f32x4 *>(ps + c0), s1 = *reinterpret_cast<const f32x4 *>(ps + c0 + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4 *>(pb + c0), b1 = *reinterpret_cast<const f32x4 *>(pb + c0 + 4);
            // B operands of (offset u, column group cg): [GB + 1][3 pieces] x 16 B, two sets
            uint4 wb[2][(GB + 1) * 3];
            auto load_b = [&](int u, int cg, uint4 (&dst)[(GB + 1) * 3]) {
                const int kk = min(min(k0 + u, k_hi - 1), p.K - 1);
                const int first = cg * GB + (cg < GR ? cg : GR);
                const uint4 *w = img + ((size_t)(kk * NC32 + c32) * p.NBtot + nb0 + first) * 3 * 64 + l;
#pragma unroll
                for (int i = 0; i < (GB + 1) * 3; i++) dst[i] = w[min(i, (NBT - first) * 3 - 1) * 64];
            };
            load_b(0, 0, wb[0]);
#pragma unroll
            for (int u = 0; u < OGB; u++) {
                bf16x8 a0, a1, a2;
                if (any[u])
                    act_split8(lo[u], hi[u], p.pre_scale != nullptr, s0, s1, b0, b1, p.pre_relu ? 0.f : -INFINITY,
                               ~(idx[u] >> 31) & c_keep, a0, a1, a2);
#pragma unroll
                for (int cg = 0; cg < NG; cg++) {
                    constexpr int dummy = 0; (void)dummy;
                    const int stage = u * NG + cg;                       // compile-time after unrolling
                    // next stage's operands first
                    if (cg + 1 < NG) load_b(u, cg + 1, wb[(stage + 1) & 1]);
                    else if (u + 1 < OGB) load_b(u + 1, 0, wb[(stage + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (any[u]) {
                        const int first = cg * GB + (cg < GR ? cg : GR), count = GB + (cg < GR ? 1 : 0);
#pragma unroll
                        for (int i = 0; i < GB + 1; i++) {
                            if (i >= count) continue;
                            const uint4 &r0 = wb[stage & 1][i * 3 + 0], &r1 = wb[stage & 1][i * 3 + 1], &r2 = wb[stage & 1][i * 3 + 2];
                            const bf16x8 w0 = *reinterpret_cast<const bf16x8 *>(&r0), w1 = *reinterpret_cast<const bf16x8 *>(&r1),
                                         w2 = *reinterpret_cast<const bf16x8 *>(&r2);
                            f32x4 &d = acc[first + i];
                            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, w0, d, 0, 0, 0);
                            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, w2, d, 0, 0, 0);
                            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, w1, d, 0, 0, 0);
                            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, w0, d, 0, 0, 0);
                            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, w1, d, 0, 0, 0);
                            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, w0, d, 0, 0, 0);
                        }
                    }
                }
            }
        }
    }
    if (wave_id() > 0) {
#pragma unroll
        for (int nb = 0; nb < NBT; nb++)
#pragma unroll
            for (int r = 0; r < 4; r++) s_acc[((size_t)(wave_id() - 1) * NBT + nb) * 256 + r * 64 + l] = acc[nb][r];
    }
    __syncthreads();
    if (wave_id() == 0) {
        for (int w = 1; w < waves; w++)
#pragma unroll
            for (int nb = 0; nb < NBT; nb++)
#pragma unroll
                for (int r = 0; r < 4; r++) acc[nb][r] += s_acc[((size_t)(w - 1) * NBT + nb) * 256 + r * 64 + l];
        store_tile<NBT>(p, row0, nb0, acc, s_part);
    }
    if (with_partial) stats_flush<NBT>(p, s_part, 1, nb0);
}

// ---- the same with RT row tiles per block (levels of ~3k - 17k rows) ------------------------------------------
// With one 16-row tile per block every tile streams the whole weight image out of L2 (128 -> 128 at 12k rows: 754
// blocks x 2.6 MB = 2 GB per launch, ~160 us at the L2's rate for 25 us of matrix work).  Here a block owns RT
// consecutive tiles: a wave walks its share of the offsets one (offset, 32-channel chunk) step at a time, loads that
// step's B operands ONCE (two column groups, double-buffered in registers: the next group's / next step's loads fly
// during this group's MFMAs) and uses them for all RT tiles; the next step's gathers and the table entries of the offset
// after are in flight during the step.  Partial accumulators of the waves meet in LDS; wave t finishes tile t.
template <int NBT, int RT>
__global__ __launch_bounds__(256) void spconv_fwd_small_bf3_rt_kernel(ConvArgs p)
{
    extern __shared__ float lds[];
    constexpr int CG = (NBT + 1) / 2;              // column blocks per operand group; always two groups
    const int l = lane_id(), g8 = l >> 4;
    const int waves = blockDim.x >> 6, w = wave_id();
    const int nb0 = blockIdx.y * NBT;
    const int NC32 = (p.Cin + 31) >> 5;
    float *s_acc = lds;                                             // [waves][RT][NBT][4][64]
    float *s_part = lds + (size_t)waves * RT * NBT * 256;          // [waves] statistics slots when bn_x / out_stats
    const bool with_partial = p.bn_x != nullptr || p.out_stats != 0;
    if (with_partial) stats_clear<NBT>(s_part, waves);  // wave t finishes tile t: a slot per wave
    const int tile0 = blockIdx.x * RT;
    int safe_row[RT];
    bool row_ok[RT];
#pragma unroll
    for (int t = 0; t < RT; t++) {
        const int r = (tile0 + t) * 16 + (l & 15);
        row_ok[t] = r < p.Vout;
        safe_row[t] = row_ok[t] ? r : 0;
    }
    f32x4 acc[RT][NBT];
#pragma unroll
    for (int t = 0; t < RT; t++)
#pragma unroll
        for (int nb = 0; nb < NBT; nb++) acc[t][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const uint4 *img = reinterpret_cast<const uint4 *>(p.wfb);
    const int ogw = (p.K + waves - 1) / waves;                     // offsets per wave
    const int k_lo = w * ogw, k_hi = min(p.K, k_lo + ogw);
    const int nsteps = max(0, k_hi - k_lo) * NC32;                 // step = (offset, 32-channel chunk)
    const bool affine = p.pre_scale != nullptr;
    const float relu_floor = p.pre_relu ? 0.f : -INFINITY;

    auto load_idx = [&](int k, int (&idx)[RT]) {
        const bool valid = k < k_hi;
        const int kk = min(k, p.K - 1);
#pragma unroll
        for (int t = 0; t < RT; t++) {
            const int v = p.nbr[(size_t)kk * p.Vout + safe_row[t]];
            idx[t] = v | ((row_ok[t] && valid) ? 0 : -1);
        }
    };
    struct Rows { f32x4 lo[RT], hi[RT]; };
    auto gather = [&](int c32, const int (&idx)[RT], Rows &g) {
        const int cr = 32 * c32 + 8 * g8;
        const int c0 = cr < p.Cin ? cr : p.Cin - 8;
#pragma unroll
        for (int t = 0; t < RT; t++) {
            const float *row = p.in + (size_t)max(idx[t], 0) * p.Cin + c0;
            g.lo[t] = *reinterpret_cast<const f32x4 *>(row);
            g.hi[t] = *reinterpret_cast<const f32x4 *>(row + 4);
        }
    };
    auto load_b = [&](int k, int c32, int cg, uint4 (&dst)[CG * 3]) {
        const int kk = min(k, p.K - 1), first = cg * CG;
        const uint4 *src = img + ((size_t)(kk * NC32 + c32) * p.NBtot + nb0 + first) * 3 * 64 + l;
#pragma unroll
        for (int i = 0; i < CG * 3; i++) dst[i] = src[min(i, (NBT - first) * 3 - 1) * 64];
    };

    int idx_c[RT], idx_n[RT];
    Rows g_c;
    uint4 wb[2][CG * 3];
    load_idx(k_lo, idx_c);
    gather(0, idx_c, g_c);
    load_idx(k_lo + 1, idx_n);
    load_b(k_lo, 0, 0, wb[0]);
    int k = k_lo, c32 = 0;
    for (int s = 0; s < nsteps; s++) {
        const bool wrap = c32 + 1 == NC32;
        const int c32n = wrap ? 0 : c32 + 1, kn = wrap ? k + 1 : k;
        // next step's rows (clamped row 0 behind the last step)
        Rows g_n;
        if (wrap) gather(0, idx_n, g_n);
        else gather(c32n, idx_c, g_n);
        const int cr = 32 * c32 + 8 * g8;
        const int c_keep = cr < p.Cin ? -1 : 0;
        const int c0 = cr < p.Cin ? cr : p.Cin - 8;
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, b0 = s0, b1 = s0;
        if (affine) {
            s0 = *reinterpret_cast<const f32x4 *>(p.pre_scale + c0); s1 = *reinterpret_cast<const f32x4 *>(p.pre_scale + c0 + 4);
            b0 = *reinterpret_cast<const f32x4 *>(p.pre_shift + c0); b1 = *reinterpret_cast<const f32x4 *>(p.pre_shift + c0 + 4);
        }
        bf16x8 a[RT][3];
        bool any[RT];
#pragma unroll
        for (int t = 0; t < RT; t++) {
            any[t] = __ballot(idx_c[t] >= 0) != 0ull;
            if (any[t])
                act_split8(g_c.lo[t], g_c.hi[t], affine, s0, s1, b0, b1, relu_floor, ~(idx_c[t] >> 31) & c_keep, a[t][0], a[t][1],
                           a[t][2]);
        }
#pragma unroll
        for (int cg = 0; cg < 2; cg++) {
            // the other operand set: this step's second column group, or the next step's first
            if (cg == 0) load_b(k, c32, 1, wb[1]);
            else load_b(kn, c32n, 0, wb[0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < CG; i++) {
                const int nb = cg * CG + i;
                if (nb >= NBT) continue;
                const bf16x8 w0 = *reinterpret_cast<const bf16x8 *>(&wb[cg][i * 3 + 0]), w1 = *reinterpret_cast<const bf16x8 *>(&wb[cg][i * 3 + 1]),
                             w2 = *reinterpret_cast<const bf16x8 *>(&wb[cg][i * 3 + 2]);
#pragma unroll
                for (int t = 0; t < RT; t++) {
                    if (!any[t]) continue;
                    f32x4 &d = acc[t][nb];
                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t][2], w0, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t][0], w2, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t][1], w1, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t][1], w0, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t][0], w1, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t][0], w0, d, 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < RT; t++) { g_c.lo[t] = g_n.lo[t]; g_c.hi[t] = g_n.hi[t]; }
        if (wrap) {
#pragma unroll
            for (int t = 0; t < RT; t++) idx_c[t] = idx_n[t];
            load_idx(kn + 1, idx_n);
        }
        k = kn; c32 = c32n;
    }
#pragma unroll
    for (int t = 0; t < RT; t++)
#pragma unroll
        for (int nb = 0; nb < NBT; nb++)
#pragma unroll
            for (int r = 0; r < 4; r++) s_acc[(((size_t)w * RT + t) * NBT + nb) * 256 + r * 64 + l] = acc[t][nb][r];
    __syncthreads();
    for (int t = w; t < RT; t += waves) {
        if (tile0 + t >= p.ntiles) break;
        f32x4 sum[NBT];
#pragma unroll
        for (int nb = 0; nb < NBT; nb++) sum[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int ww = 0; ww < waves; ww++)
#pragma unroll
            for (int nb = 0; nb < NBT; nb++)
#pragma unroll
                for (int r = 0; r < 4; r++) sum[nb][r] += s_acc[(((size_t)ww * RT + t) * NBT + nb) * 256 + r * 64 + l];
        store_tile<NBT>(p, (tile0 + t) * 16, nb0, sum, s_part + (size_t)w * stat_slot_floats(NBT));
    }
    if (with_partial) stats_flush<NBT>(p, s_part, waves, nb0);
}

template <int NBT, int RT>
int launch_fwd_small_bf3_rt(const ConvArgs &p, dim3 grid, int threads, size_t lds, hipStream_t stream)
{
    static const hipError_t attr = raise_lds_ceiling((const void *)spconv_fwd_small_bf3_rt_kernel<NBT, RT>);
    MS3D_CHECK(attr);
    spconv_fwd_small_bf3_rt_kernel<NBT, RT><<<grid, threads, lds, stream>>>(p);
    MS3D_LAUNCH_CHECK();
    return 0;
}

template <int NBT>
int launch_fwd_small_bf3(const ConvArgs &p, dim3 grid, int threads, size_t lds, hipStream_t stream)
{
    spconv_fwd_small_bf3_kernel<NBT><<<grid, threads, lds, stream>>>(p);
    MS3D_LAUNCH_CHECK();
    return 0;
}

template <int NBT>
int launch_fwd_small(const ConvArgs &p, dim3 grid, int threads, size_t lds, bool aligned, hipStream_t stream)
{
    // The geometry may have been laid out for the bf16x3 three-tile kernel (4 waves x RT x nbt KB of LDS, 72-96+ KB at
    // >= 96 output channels) while this call has no bf16x3 image (the exact-f32 entry point, a BatchNorm without ReLU,
    // aux kind 1): this kernel parks (waves - 1) x nbt accumulator tiles + the statistics only -- ask for that, not for
    // the other kernel's footprint (which would exceed the 64 KB a kernel gets without raising its ceiling)
    const size_t need = ((size_t)(threads / 64 - 1) * NBT * 256 + ((p.bn_x || p.out_stats) ? stat_slot_floats(NBT) : 0)) * sizeof(float) + 16;
    if (lds > need) lds = need;
    if (aligned)
        spconv_fwd_small_kernel<NBT, true><<<grid, threads, lds, stream>>>(p);
    else
        spconv_fwd_small_kernel<NBT, false><<<grid, threads, lds, stream>>>(p);
    MS3D_LAUNCH_CHECK();
    return 0;
}

template <int NBT>
int launch_fwd(const ConvArgs &p, dim3 grid, int threads, size_t lds, bool aligned, hipStream_t stream)
{
    // the dynamic-LDS ceiling of a kernel is raised ONCE, to the most any launch may ask for (per launch it would be a
    // race between host threads that use different sizes)
    const bool res = p.G >= p.K;
#define MS3D_FWD_LAUNCH(A, R)                                                                                        \
    do {                                                                                                             \
        static const hipError_t attr = raise_lds_ceiling((const void *)spconv_fwd_kernel<NBT, A, R>);                \
        MS3D_CHECK(attr);                                                                                            \
        spconv_fwd_kernel<NBT, A, R><<<grid, threads, lds, stream>>>(p);                                             \
    } while (0)
    if (aligned && res) MS3D_FWD_LAUNCH(true, true);
    else if (aligned) MS3D_FWD_LAUNCH(true, false);
    else if (res) MS3D_FWD_LAUNCH(false, true);
    else MS3D_FWD_LAUNCH(false, false);
#undef MS3D_FWD_LAUNCH
    MS3D_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ weight-stationary kernels of the coarse levels (round 6)
// Levels with <= ~3k output rows (L4-L6 of the backbone: 2.5k / 500 / 112 rows at 80..224 channels, and the 2c -> c layers
// behind their concatenations).  The one-tile kernels above hand every 16-row tile the whole column slice of the weight
// image: 160 -> 160 at 2.6k rows streams 162 x 2.8 MB through L2 for 3 MB of rows, and a launch is a chain of ~45 dependent
// L2 round trips per wave.  Here the WEIGHTS stay put:
//   block  = (offset group g of kg offsets, column slice c of NBT blocks, row part r of tpp tiles); its slab
//            W[k_lo..k_lo+kg)[all Cin][NBT*16] (MS3D_WS_LDS_KB: 128 KB and 16 waves by default, measured best; the first build
//            used 64 KB and two 4-wave blocks per CU) is copied into LDS ONCE and every tile of the part runs through it --
//            weights leave L2 (row parts) x |W| per launch instead of (tiles) x |W|;
//   wave   = tiles w, w + waves, .. of the part; per tile the (offset, 16-channel chunk) items of the group are walked in ONE flat
//            sequence, WS_RB row gathers in flight and the next WS_RB behind them (two register sets): a tile costs two
//            dependent round trips (table entries -- fetched a tile ahead -- and rows) whatever Cin is; B fragments are
//            conflict-free ds_read_b32 rows of the slab; offsets no row of the tile has skip their MFMAs;
//   split-K: the ng = ceil(K / kg) groups of a unit (c, r) leave their partial tiles in a slab area (16 B per lane,
//            accumulator layout) and draw a ticket; the block that draws the LAST ticket adds the ng partials of every tile
//            in GROUP order (not arrival order: bit-reproducible), and runs the usual epilogue (residual / bias / statistics /
//            fused BatchNorm-backward mask and sums).  Hand-off as the CDNA4 guide prescribes for an in-launch split-K: plain
//            16-byte stores -> per-wave vmcnt(0) -> barrier -> ONE lane: agent-scope release fence, vmcnt(0), relaxed agent
//            fetch_add; the last arriver: ONE agent-scope acquire -> barrier -> plain loads.  Correct for any placement of a
//            unit's blocks; for speed the block id puts them on one XCD (id & 7 = unit & 7) and next to each other in dispatch
//            order.  The last arriver zeroes the counter for the next launch on the stream.
// (WS_RB row gathers in flight per register set: 8 in the <= 8-wave form, 4 in the 16-wave form whose waves have 128 registers)

struct WsBlock { int u, g, nb0, k_lo, kcnt, t_begin, t_end; };

template <int NBT>
__device__ __forceinline__ bool ws_block(const ConvArgs &p, WsBlock &b)
{
    const int ny = p.NBtot / NBT, units = p.ws_R * ny;
    const int id = (int)blockIdx.x;
    b.u = (id / (8 * p.ws_ng)) * 8 + (id & 7);
    b.g = (id >> 3) % p.ws_ng;
    if (b.u >= units) return false;
    const int c = b.u % ny, r = b.u / ny;
    b.nb0 = c * NBT;
    b.k_lo = b.g * p.ws_kg;
    b.kcnt = min(p.K, b.k_lo + p.ws_kg) - b.k_lo;
    b.t_begin = r * p.ws_tpp;
    b.t_end = min(p.ntiles, b.t_begin + p.ws_tpp);
    return true;
}

__device__ __forceinline__ f32x4 *ws_slab_ptr(const ConvArgs &p, int g, int tile, int nb)
{
    return reinterpret_cast<f32x4 *>(p.ws_slabs) + (((size_t)g * p.ntiles + tile) * p.NBtot + nb) * 64 + lane_id();
}

// after the block's tile loop (ng > 1): publish, draw the ticket, and -- last arriver -- combine + epilogue
template <int NBT>
__device__ __forceinline__ void ws_finish(const ConvArgs &p, const WsBlock &b, float *s_part, int *s_flag)
{
    const int w = wave_id(), waves = blockDim.x >> 6;
    const bool with_partial = p.bn_x != nullptr || p.out_stats != 0;
    if (p.ws_ng > 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned t = __hip_atomic_fetch_add(&p.ws_cnt[b.u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = t == (unsigned)(p.ws_ng - 1);
            if (last) {
                __hip_atomic_store(&p.ws_cnt[b.u], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the next launch on this stream
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            *s_flag = last;
        }
        __syncthreads();
        if (!*s_flag) return;
        float *slot = s_part + (size_t)w * stat_slot_floats(NBT);
        for (int tile = b.t_begin + w; tile < b.t_end; tile += waves) {
            // GB groups' partials requested together (unconditional, clamped), added in group order
            constexpr int GB = NBT <= 2 ? 8 : 4;
            f32x4 sum[NBT];
#pragma unroll
            for (int nb = 0; nb < NBT; nb++) sum[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int g0 = 0; g0 < p.ws_ng; g0 += GB) {
                f32x4 v[GB][NBT];
#pragma unroll
                for (int j = 0; j < GB; j++)
#pragma unroll
                    for (int nb = 0; nb < NBT; nb++) v[j][nb] = *ws_slab_ptr(p, min(g0 + j, p.ws_ng - 1), tile, b.nb0 + nb);
#pragma unroll
                for (int j = 0; j < GB; j++) {
                    if (g0 + j >= p.ws_ng) break;
#pragma unroll
                    for (int nb = 0; nb < NBT; nb++) sum[nb] += v[j][nb];
                }
            }
            store_tile<NBT>(p, tile * 16, b.nb0, sum, slot);
        }
    }
    if (with_partial) {
        // one statistics row per unit (ms3d_spconv_partial_blocks counts R x ny): the slots in wave order, other slices' columns zero
        __syncthreads();
        float *dst = p.bn_partial + (size_t)b.u * 2 * p.Cout;
        for (int t = threadIdx.x; t < 2 * p.Cout; t += blockDim.x) {
            const int which = t >= p.Cout, c = t - which * p.Cout - 16 * b.nb0;
            float sum = 0.f;
            if (c >= 0 && c < NBT * 16)
                for (int ww = 0; ww < waves; ww++) sum += s_part[(size_t)ww * stat_slot_floats(NBT) + which * NBT * 16 + c];
            dst[t] = sum;
        }
    }
}

template <int NBT, int WS_RB, int MAXT>
__global__ __launch_bounds__(MAXT) void spconv_fwd_ws_kernel(ConvArgs p)
{
    extern __shared__ float lds[];
    WsBlock b;
    if (!ws_block<NBT>(p, b)) return;
    const int l = lane_id(), q = l >> 4, il = l & 15, w = wave_id(), waves = blockDim.x >> 6;
    // LDS: weights [kg][NCH][4][NBT][64] | scale[16 NCH] | shift[16 NCH] | table entries [waves][kg][16] | statistics slots | flag
    float *sW = lds;
    float *s_scale = sW + (size_t)p.ws_kg * p.NCH * 4 * NBT * 64;
    float *s_shift = s_scale + p.NCH * 16;
    int *s_idx = reinterpret_cast<int *>(s_shift + p.NCH * 16);
    float *s_part = reinterpret_cast<float *>(s_idx + waves * p.ws_kg * 16);
    int *s_flag = reinterpret_cast<int *>(s_part + (size_t)waves * stat_slot_floats(NBT));
    {
        const int rows = b.kcnt * p.NCH * 4, slab4 = NBT * 16;
        const float *src = p.wf + ((size_t)b.k_lo * p.NCH * 4 * p.NBtot + b.nb0) * 64;
        for (int e = threadIdx.x; e < rows * slab4; e += blockDim.x) {
            const int r = e / slab4, c4 = e - r * slab4;
            reinterpret_cast<f32x4 *>(sW)[e] = reinterpret_cast<const f32x4 *>(src + (size_t)r * p.NBtot * 64)[c4];
        }
        if (p.pre_scale)
            for (int e = threadIdx.x; e < p.Cin; e += blockDim.x) { s_scale[e] = p.pre_scale[e]; s_shift[e] = p.pre_shift[e]; }
    }
    const bool with_partial = p.bn_x != nullptr || p.out_stats != 0;
    if (with_partial) stats_clear<NBT>(s_part, waves);
    __syncthreads();
    const bool affine = p.pre_scale != nullptr;
    int *my_idx = s_idx + w * p.ws_kg * 16;
    const int nitems = b.kcnt * p.NCH;
    constexpr int NI = 3;                               // table entries per lane: offsets q, q + 4, q + 8 (kg <= 12)
    auto load_entries = [&](int tile, int (&e)[NI]) {
        const int row = tile * 16 + il;
#pragma unroll
        for (int j = 0; j < NI; j++) {
            const int uu = q + 4 * j;
            const bool ok = uu < b.kcnt && row < p.Vout && tile < b.t_end;
            const int v = p.nbr[(size_t)(b.k_lo + (uu < b.kcnt ? uu : 0)) * p.Vout + (row < p.Vout ? row : 0)];
            e[j] = v | (ok ? 0 : -1);
        }
    };
    int ent[NI];
    load_entries(b.t_begin + w, ent);
    for (int tile = b.t_begin + w; tile < b.t_end; tile += waves) {
#pragma unroll
        for (int j = 0; j < NI; j++)
            if (q + 4 * j < p.ws_kg) my_idx[(q + 4 * j) * 16 + il] = ent[j];
        load_entries(tile + waves, ent);                // the next tile's entries travel under this tile's work
        f32x4 acc[NBT];
#pragma unroll
        for (int nb = 0; nb < NBT; nb++) acc[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // item = (offset uu of the group, 16-channel chunk ch), walked as one flat sequence; (fu, fc) = the fetch cursor,
        // (cu, cc) = the compute cursor
        int fu = 0, fc = 0, cu = 0, cc = 0;
        auto fetch = [&](int i0, f32x4 (&a)[WS_RB], int (&ri)[WS_RB]) {
#pragma unroll
            for (int j = 0; j < WS_RB; j++) {
                const bool live = i0 + j < nitems;
                const int r = my_idx[min(fu, b.kcnt - 1) * 16 + il] | (live ? 0 : -1);
                ri[j] = r;
                a[j] = *reinterpret_cast<const f32x4 *>(p.in + (size_t)max(r, 0) * p.Cin + 16 * fc + 4 * q);
                if (live) { fc++; if (fc == p.NCH) { fc = 0; fu++; } }
            }
        };
        auto compute = [&](int i0, f32x4 (&a)[WS_RB], const int (&ri)[WS_RB]) {
#pragma unroll
            for (int j = 0; j < WS_RB; j++) {
                if (i0 + j >= nitems) break;
                const bool any = __ballot(ri[j] >= 0) != 0ull;
                if (any) {
                    f32x4 v = a[j];
                    if (affine) {
                        const f32x4 s = *reinterpret_cast<const f32x4 *>(s_scale + 16 * cc + 4 * q);
                        const f32x4 sh = *reinterpret_cast<const f32x4 *>(s_shift + 16 * cc + 4 * q);
#pragma unroll
                        for (int t = 0; t < 4; t++) {
                            const float z = fmaf(v[t], s[t], sh[t]);
                            v[t] = p.pre_relu ? fmaxf(z, 0.f) : z;
                        }
                    }
                    const int keep = ~(ri[j] >> 31);
#pragma unroll
                    for (int t = 0; t < 4; t++) v[t] = __int_as_float(__float_as_int(v[t]) & keep);
                    const float *wp = sW + (size_t)((cu * p.NCH + cc) * 4) * NBT * 64 + l;
#pragma unroll
                    for (int t = 0; t < 4; t++)
#pragma unroll
                        for (int nb = 0; nb < NBT; nb++)
                            acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[t], wp[(t * NBT + nb) * 64], acc[nb], 0, 0, 0);
                }
                cc++; if (cc == p.NCH) { cc = 0; cu++; }
            }
        };
        f32x4 a0[WS_RB], a1[WS_RB];
        int r0[WS_RB], r1[WS_RB];
        fetch(0, a0, r0);
        for (int i0 = 0; i0 < nitems; i0 += 2 * WS_RB) {
            fetch(i0 + WS_RB, a1, r1);
            compute(i0, a0, r0);
            fetch(i0 + 2 * WS_RB, a0, r0);
            compute(i0 + WS_RB, a1, r1);
        }
        if (p.ws_ng == 1) {
            store_tile<NBT>(p, tile * 16, b.nb0, acc, s_part + (size_t)w * stat_slot_floats(NBT));
        } else {
#pragma unroll
            for (int nb = 0; nb < NBT; nb++) *ws_slab_ptr(p, b.g, tile, b.nb0 + nb) = acc[nb];
        }
    }
    ws_finish<NBT>(p, b, s_part, s_flag);
}

template <int NBT>
int launch_fwd_ws(const ConvArgs &p, int nblk, int threads, size_t lds, hipStream_t stream)
{
    if (threads > 512) {
        static const hipError_t attr = raise_lds_ceiling((const void *)spconv_fwd_ws_kernel<NBT, 4, 1024>);
        MS3D_CHECK(attr);
        spconv_fwd_ws_kernel<NBT, 4, 1024><<<nblk, threads, lds, stream>>>(p);
    } else {
        static const hipError_t attr = raise_lds_ceiling((const void *)spconv_fwd_ws_kernel<NBT, 8, 512>);
        MS3D_CHECK(attr);
        spconv_fwd_ws_kernel<NBT, 8, 512><<<nblk, threads, lds, stream>>>(p);
    }
    MS3D_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ pair-list forward / backward-data
// Full-resolution levels are sparse: a voxel of a 2 cm ScanNet-like surface has ~3-9 of its 27 neighbours, so a
// 16-row output tile that runs one MFMA group per offset whenever ANY of its rows has that neighbour keeps the matrix
// pipe ~15 % useful, gathers 27 x 16 rows (most of them clamped dummies) and is bound by instruction issue.
// With a pair list (ms3d_kmap_pairlist_build: per 64-row tile the valid pairs grouped by offset in 16-pair batches)
// a wave owns a 64-row tile and per batch issues
//     1 entry load, 1 row gather (16 B / lane), 1 weight-fragment read (ds_read_b128), 4 MFMAs per (ch, nb) with the
//     roles swapped -- D^T[cout][pair] = W_k^T[cout][cin] x in^T[cin][pair] -- so that a lane ends up with 4
//     consecutive output channels of ITS pair, and one 16-B read-modify-write of the wave's LDS accumulator tile.
// Batches run in ascending offset order and a batch holds distinct output rows (pads go to a dummy row), so the
// per-output summation order is fixed: deterministic, no atomics.  The epilogue streams the tile out row-major
// (residual / bias / output statistics / fused BN-backward mask and sums exactly as store_tile).
// CR = rows per tile of the list: 64 (ms3d_kmap_pairlist_build), or 32 for 32 -> 32 layers (round 6): their weight image is
// 110 KB of LDS, and with 8.3 KB accumulator tiles only four waves fit beside it, so rounds 1-5 ran them as two 16-column
// slices of 12 waves -- every row gathered twice, twice the address / BatchNorm / entry work per MFMA (round-6 counters: the
// kernel is bound by instruction issue, not bytes).  With 32-row tiles (4.2 KB) nine waves hold both column blocks.
constexpr int PL_ROWS_NARROW = 32;

__host__ __device__ constexpr size_t pairlist_wave_floats(int nbt, int cr = MS3D_PL_ROWS) { return (size_t)(cr + 1) * nbt * 16; }

// (the narrow form never runs more than 10 waves -- 110 KB of weights + 10 x 4.2 KB -- so it is compiled for 640 threads:
// 168 registers instead of 128, no scratch in the straight-line batch loop)
template <int NBT, int NCH, int CR = MS3D_PL_ROWS>
__global__ __launch_bounds__(CR == PL_ROWS_NARROW ? 640 : 1024) void spconv_fwd_pairlist_kernel(ConvArgs p)
{
    struct { const float *bias, *bn_scale, *bn_shift, *bn_mean, *bn_invstd; } ep = {p.bias, p.bn_scale, p.bn_shift, p.bn_mean, p.bn_invstd};
    extern __shared__ float lds[];
    constexpr int CW = NBT * 16;  // accumulator row width
    constexpr int F4 = CW / 4;
    constexpr int CBU = NCH == 1 ? 8 : 4;  // batches whose gathers are in flight together
    const int l = lane_id(), q = l >> 4, jl = l & 15;
    const int waves = blockDim.x >> 6;
    const int nb0 = blockIdx.y * NBT;
    // Accumulator tile: row-major, but the 16-byte chunk c of row r sits at chunk c ^ ((r / RPL) % F4), RPL = rows per
    // 256 bytes.  The pairs of a batch hit (mostly neighbouring) rows with the SAME chunk index: unswizzled, rows that
    // are RPL apart share their 4 banks -- 16 lanes on 4 (NBT = 1) or 2 (NBT = 2) bank groups, a 4..8-way conflict on
    // every read-modify-write, which is what bounded this kernel.  Swizzled, 16 consecutive rows cover all 64 banks.
    auto acc_slot = [](int r, int c) { return r * CW + 4 * (c ^ ((r / (16 / F4)) & (F4 - 1))); };
    f32x4 *sW4 = reinterpret_cast<f32x4 *>(lds);  // [(k*NCH + ch)*NBT + nb][lane] -> the 4 k-steps t of that lane
    const int wslots = p.K * NCH * NBT * 64;
    float *s_part = lds + (size_t)wslots * 4;  // [2*Cout]
    int *s_next = reinterpret_cast<int *>(s_part + ((2 * p.Cout + 3) & ~3));  // tile pick counter (+ 3 pad: 16-B alignment)
    float *acc_t = s_part + ((2 * p.Cout + 3) & ~3) + 4 + (size_t)wave_id() * pairlist_wave_floats(NBT, CR);  // [(CR+1)][CW]

    const int nblk = gridDim.x;
    const int per_xcd = (nblk + 7) / 8;
    int vb = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;  // blocks of one XCD take neighbouring tiles
    if (nblk % 8 != 0) vb = blockIdx.x;
    // one part of the list per block (near-equal batch counts, see pairlist_parts_kernel); the block's waves pick the
    // part's tiles longest first off an LDS counter (the first `waves` picks are static).  A pick is one 16-byte
    // descriptor (tile, first batch, end batch); a wave's first descriptor and first group of entries are requested
    // here, in front of the weight staging, and every later tile's during the tile before it: order -> tile_start ->
    // entries -> gather were four dependent round trips at the start of every tile.
    const int *__restrict__ part_start = p.pl_tile_start + p.ntiles + 1;  // p.ntiles counts CR-row tiles here
    const int4 *__restrict__ picks = reinterpret_cast<const int4 *>(p.pl_tile_start + MS3D_PL_SCHED_OFFSET(p.ntiles));
    const int2 *__restrict__ entries = reinterpret_cast<const int2 *>(p.pl_entries);
    constexpr int EH = CBU / 4;  // 64-pair halves per group of CBU batches
    const int t0 = part_start[vb], nmine = part_start[vb + 1] - t0;
    int slot = wave_id();
    auto first_entries = [&](const int4 &d, int2 (&e)[EH]) {
        const int last = max(d.z - d.y - 1, 0);
#pragma unroll
        for (int h = 0; h < EH; h++)
            e[h] = entries[(size_t)d.y * 16 + (unsigned)(min(4 * h + (l >> 4), last) * 16 + jl)];
    };
    int4 desc = picks[min(t0 + min(slot, max(nmine - 1, 0)), p.ntiles - 1)];  // (an empty part: any valid pick, never used)
    int2 nxt[EH];
    first_entries(desc, nxt);

    // weights of this block's column slice: global fragment order [k][ch][t][nb][lane] -> LDS [k][ch][nb][lane][t].
    // A thread fetches 16 bytes (lanes 4g..4g+3 of step t), the four threads of a quad (t = 0..3) transpose their 4x4
    // block with two exchange stages, and each stores the 16 bytes of ONE lane's four steps: 4x fewer global loads
    // than one dword per (lane, step) -- the staging was 4-8 us of every launch (27 ... 55 KB per workgroup).
    for (int e = threadIdx.x; e < wslots; e += blockDim.x) {
        const int t = e & 3, g = (e >> 2) & 15, r = e >> 6;
        const int nb = r % NBT, kc = r / NBT;  // kc = k*NCH + ch
        f32x4 a = *reinterpret_cast<const f32x4 *>(p.wf + ((size_t)(kc * 4 + t) * p.NBtot + nb0 + nb) * 64 + 4 * g);
        {
            const bool odd = t & 1;
            const float rx = __shfl_xor(odd ? a[0] : a[1], 1, 64), ry = __shfl_xor(odd ? a[2] : a[3], 1, 64);
            if (odd) { a[0] = rx; a[2] = ry; } else { a[1] = rx; a[3] = ry; }
        }
        {
            const bool hi = t & 2;
            const float rx = __shfl_xor(hi ? a[0] : a[2], 2, 64), ry = __shfl_xor(hi ? a[1] : a[3], 2, 64);
            if (hi) { a[0] = rx; a[1] = ry; } else { a[2] = rx; a[3] = ry; }
        }
        sW4[e] = a;  // e = r*64 + 4g + t: lane 4g + t of fragment r, steps 0..3
    }
    const bool with_partial = p.bn_x != nullptr || p.out_stats != 0;
    if (threadIdx.x == 0) *s_next = waves;
    for (int e = l; e < (CR + 1) * F4; e += 64) reinterpret_cast<f32x4 *>(acc_t)[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    f32x4 st1 = {0.f, 0.f, 0.f, 0.f}, st2 = {0.f, 0.f, 0.f, 0.f};  // per-lane column sums (columns 4*(l % F4)..+3)
    const int c4 = l % F4, col = 16 * nb0 + 4 * c4;
    // fused input BatchNorm(+ReLU): this lane's 4 channels per chunk, loaded once (inside the batch loop the two loads
    // were re-issued and waited for in every chunk)
    f32x4 pre_sc[NCH], pre_sh[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) {
        pre_sc[ch] = p.pre_scale ? *reinterpret_cast<const f32x4 *>(p.pre_scale + 16 * ch + 4 * q) : (f32x4){1.f, 1.f, 1.f, 1.f};
        pre_sh[ch] = p.pre_scale ? *reinterpret_cast<const f32x4 *>(p.pre_shift + 16 * ch + 4 * q) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    int round = 0;
    while (slot < nmine) {
        // The wave's next tile (its descriptor is requested now).  Round 5: a FIXED schedule -- the part's tiles are listed
        // longest first; round r hands tiles r*W .. r*W + W - 1 to the waves in alternating direction (the wave that got the
        // shortest tile of a round gets the longest of the next: what the LDS pick counter of rounds 1-4 did whenever a tile's
        // time followed its batch count), so the tiles a wave sums its statistics over, and their order, no longer depend on
        // timing: the epilogue statistics are bit-reproducible.  MS3D_PL_DYNAMIC=1 restores the counter (A/B measurements).
        ++round;
        int next_slot = round * waves + ((round & 1) ? waves - 1 - wave_id() : wave_id());
        if (p.dyn_picks) {
            int pick = 0;
            if (l == 0) pick = atomicAdd(s_next, 1);
            next_slot = pick;
        }
        const int tile = __builtin_amdgcn_readfirstlane(desc.x);
        const int row0 = tile * CR;
        const int b_begin = __builtin_amdgcn_readfirstlane(desc.y);
        const int b_end = __builtin_amdgcn_readfirstlane(desc.z);
        const int2 *__restrict__ tile_entries = entries + (size_t)b_begin * 16;
        // Entries of a group of CBU batches (16 * CBU pairs): lane L holds pairs L and 64 + L, one coalesced 8-byte load
        // each, and a batch's 16 entries reach the four quads that need them through the LDS crossbar (ds_bpermute).
        // The next group's entries are requested before this group's gathers, so the entry round trip is off the
        // wave's dependent chain (entry -> gather -> MFMA -> accumulate); a batch index past the tile re-reads its last batch.
        const int last_b = max(b_end - b_begin - 1, 0);
        next_slot = __builtin_amdgcn_readfirstlane(next_slot);
        const int4 next_desc = picks[t0 + min(next_slot, nmine - 1)];
        for (int b0 = b_begin; b0 < b_end; b0 += CBU) {
            int2 ent[CBU];
            f32x4 a[CBU][NCH];
            int2 cur[EH];
#pragma unroll
            for (int h = 0; h < EH; h++) {
                cur[h] = nxt[h];
                nxt[h] = tile_entries[(unsigned)(min(b0 - b_begin + CBU + 4 * h + (l >> 4), last_b) * 16 + jl)];
            }
#pragma unroll
            for (int u = 0; u < CBU; u++) {
                const int src = ((u & 3) * 16 + jl) << 2;
                ent[u].x = __builtin_amdgcn_ds_bpermute(src, cur[u >> 2].x);
                ent[u].y = __builtin_amdgcn_ds_bpermute(src, cur[u >> 2].y);
            }
#pragma unroll
            for (int u = 0; u < CBU; u++) {
                // wave-uniform base + 32-bit lane offset (Vin * Cin < 2^32 elements is checked by the launcher): the 64-bit
                // per-lane address arithmetic was a third of the loop's vector instructions
                const unsigned off = (unsigned)ent[u].x * (unsigned)(NCH * 16) + 4u * (unsigned)q;
#pragma unroll
                for (int ch = 0; ch < NCH; ch++) a[u][ch] = *reinterpret_cast<const f32x4 *>(p.in + off + 16 * ch);
            }
            if (p.pre_scale) {
#pragma unroll
                for (int ch = 0; ch < NCH; ch++) {
                    const f32x4 sc = pre_sc[ch], sh = pre_sh[ch];
#pragma unroll
                    for (int u = 0; u < CBU; u++)
#pragma unroll
                        for (int t = 0; t < 4; t++) {
                            const float x = fmaf(a[u][ch][t], sc[t], sh[t]);
                            a[u][ch][t] = p.pre_relu ? fmaxf(x, 0.f) : x;
                        }
                }
            }
            auto multiply_accumulate = [&](int u) {
                const int k = ent[u].y >> 8, orow = ent[u].y & 255;
                f32x4 d[NBT];
#pragma unroll
                for (int nb = 0; nb < NBT; nb++) d[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ch = 0; ch < NCH; ch++)
#pragma unroll
                    for (int nb = 0; nb < NBT; nb++) {
                        const f32x4 w = sW4[((k * NCH + ch) * NBT + nb) * 64 + l];
#pragma unroll
                        for (int t = 0; t < 4; t++)
                            d[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t], a[u][ch][t], d[nb], 0, 0, 0);
                    }
                // D^T layout: row = output channel 4q + r, column = pair jl -> 16 contiguous bytes of the pair's row.
                // Plain read-modify-write: the wave is the only writer of its tile, its LDS operations execute in
                // order, and inside one batch only pad pairs (dummy row) share an address (LDS float atomics were
                // measured at ~240 cycles per instruction).
#pragma unroll
                for (int nb = 0; nb < NBT; nb++) {
                    f32x4 *dst = reinterpret_cast<f32x4 *>(acc_t + acc_slot(orow, 4 * nb + q));
                    f32x4 cur = *dst;
#pragma unroll
                    for (int r = 0; r < 4; r++) cur[r] += d[nb][r];
                    *dst = cur;
                }
            };
            if ((NCH * NBT <= 2 || (NCH == 2 && NBT == 2)) && b0 + CBU <= b_end) {
                // Full group, one straight-line block (register budget: up to two fragments per batch).  The compiler cannot tell the weight image from the accumulator
                // tile (both LDS), so it keeps every batch's fragment read behind the previous batch's accumulator
                // write: read -> 4 dependent MFMAs -> read-modify-write, ~330 cycles per batch in sequence.  Program order
                // here is: fragments of batch u+1, accumulator rows of batch u, MFMAs of batch u, add, write.
                f32x4 wn[NCH][NBT];
#pragma unroll
                for (int ch = 0; ch < NCH; ch++)
#pragma unroll
                    for (int nb = 0; nb < NBT; nb++) wn[ch][nb] = sW4[(((ent[0].y >> 8) * NCH + ch) * NBT + nb) * 64 + l];
#pragma unroll
                for (int u = 0; u < CBU; u++) {
                    f32x4 w[NCH][NBT];
#pragma unroll
                    for (int ch = 0; ch < NCH; ch++)
#pragma unroll
                        for (int nb = 0; nb < NBT; nb++) w[ch][nb] = wn[ch][nb];
                    if (u + 1 < CBU) {
                        const int kn = ent[u + 1].y >> 8;
#pragma unroll
                        for (int ch = 0; ch < NCH; ch++)
#pragma unroll
                            for (int nb = 0; nb < NBT; nb++) wn[ch][nb] = sW4[((kn * NCH + ch) * NBT + nb) * 64 + l];
                    }
                    const int orow = ent[u].y & 255;
                    f32x4 cur[NBT], d[NBT];
#pragma unroll
                    for (int nb = 0; nb < NBT; nb++) {
                        cur[nb] = *reinterpret_cast<f32x4 *>(acc_t + acc_slot(orow, 4 * nb + q));
                        d[nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    }
                    __builtin_amdgcn_sched_barrier(0);  // the reads above are issued before the (stalling) MFMA chain, not after it
#pragma unroll
                    for (int ch = 0; ch < NCH; ch++)
#pragma unroll
                        for (int nb = 0; nb < NBT; nb++)
#pragma unroll
                            for (int t = 0; t < 4; t++)
                                d[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ch][nb][t], a[u][ch][t], d[nb], 0, 0, 0);
#pragma unroll
                    for (int nb = 0; nb < NBT; nb++) {
#pragma unroll
                        for (int r = 0; r < 4; r++) cur[nb][r] += d[nb][r];
                        *reinterpret_cast<f32x4 *>(acc_t + acc_slot(orow, 4 * nb + q)) = cur[nb];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int u = 0; u < CBU; u++) {
                    if (b0 + u >= b_end) break;
                    multiply_accumulate(u);
                }
            }
        }
        // the next tile's first entries: in flight during the epilogue (the loop above left `nxt` with a clamped re-read)
        slot = next_slot;
        desc = next_desc;
        first_entries(desc, nxt);
        __builtin_amdgcn_wave_barrier();
        // ---- epilogue: CR x CW accumulator tile, row-major, 16 B per lane per step
        // (per-tile view of the per-column operands: hoisted out of the tile loop they sat in scratch across the pair loop)
        asm volatile("" : "+s"(ep.bias), "+s"(ep.bn_scale), "+s"(ep.bn_shift), "+s"(ep.bn_mean), "+s"(ep.bn_invstd));
#pragma unroll
        for (int i = 0; i < CR * F4 / 64; i++) {
            const int r = i * (64 / F4) + l / F4;
            f32x4 *src = reinterpret_cast<f32x4 *>(acc_t + acc_slot(r, c4));
            f32x4 o4 = *src;
            *src = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int row = row0 + r;
            if (row < p.Vout) {
                const size_t o = (size_t)row * p.Cout + col;
                if (p.residual) {
                    const f32x4 rs = *reinterpret_cast<const f32x4 *>(p.residual + o);
#pragma unroll
                    for (int t = 0; t < 4; t++) o4[t] += rs[t];
                }
                if (ep.bias) {
#pragma unroll
                    for (int t = 0; t < 4; t++) o4[t] += ep.bias[col + t];
                }
                if (p.bn_x) {
                    const f32x4 x = *reinterpret_cast<const f32x4 *>(p.bn_x + o);
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        const float z = fmaf(x[t], ep.bn_scale[col + t], ep.bn_shift[col + t]);
                        const float g = (z > 0.f) ? o4[t] : 0.f;
                        o4[t] = g;
                        st1[t] += g;
                        st2[t] += g * ((x[t] - ep.bn_mean[col + t]) * ep.bn_invstd[col + t]);
                    }
                } else if (p.out_stats) {
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        st1[t] += o4[t];
                        st2[t] = fmaf(o4[t], o4[t], st2[t]);
                    }
                }
                *reinterpret_cast<f32x4 *>(p.out + o) = o4;
            }
        }
        // the dummy row collected the pad products; clear it with the rest
        if (l < F4) reinterpret_cast<f32x4 *>(acc_t + CR * CW)[l] = (f32x4){0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_wave_barrier();
    }
    if (with_partial) {
        // lanes with equal l % F4 hold the same 4 columns: fold them with shuffles, park the wave's 2 x CW sums in its
        // (now idle) accumulator tile and let 2 * Cout threads add the waves up in wave order.  (LDS float atomics from
        // every lane cost ~8 us here: 8 instructions per wave with 16 lanes per address, all waves arriving together.)
#pragma unroll
        for (int m = F4; m < 64; m <<= 1)
#pragma unroll
            for (int t = 0; t < 4; t++) {
                st1[t] += __shfl_xor(st1[t], m, 64);
                st2[t] += __shfl_xor(st2[t], m, 64);
            }
        if (l < F4) {
            *reinterpret_cast<f32x4 *>(acc_t + 4 * l) = st1;
            *reinterpret_cast<f32x4 *>(acc_t + CW + 4 * l) = st2;
        }
        __syncthreads();
        float *dst = p.bn_partial + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 * p.Cout;
        const float *wave0 = s_part + ((2 * p.Cout + 3) & ~3) + 4;
        for (int t = threadIdx.x; t < 2 * p.Cout; t += blockDim.x) {
            const int which = t >= p.Cout, c = t - which * p.Cout - 16 * nb0;  // column inside this block's slice
            float sum = 0.f;
            if (c >= 0 && c < CW)
                for (int w = 0; w < waves; w++) sum += wave0[(size_t)w * pairlist_wave_floats(NBT, CR) + which * CW + c];
            dst[t] = sum;
        }
    }
}

template <int NBT, int NCH, int CR = MS3D_PL_ROWS>
int launch_fwd_pairlist(ConvArgs p, dim3 grid, int threads, size_t lds, hipStream_t stream)
{
    p.ntiles = ms3d_divup(p.Vout, CR);
    static const hipError_t attr = raise_lds_ceiling((const void *)spconv_fwd_pairlist_kernel<NBT, NCH, CR>);
    MS3D_CHECK(attr);
    spconv_fwd_pairlist_kernel<NBT, NCH, CR><<<grid, threads, lds, stream>>>(p);
    MS3D_LAUNCH_CHECK();
    return 0;
}


// ------------------------------------------------------------------ pair-list forward / backward-data, 48+ channels
// The pair-list idea (above) for layers whose weights do not fit LDS (27 x 64 x 64 x 4 B = 442 KB).  The table walk is
// MFMA-bound there -- on work that multiplies absent neighbours (10 of 27 offsets valid per row at level 1: 2.6x the
// useful flops).  Here a wave owns a 128-row tile of the WIDE pair list (ms3d_kmap_pairlist_build_rows(.., 128, ..):
// ~3 batches per offset, 2 % pad slots) with its accumulator tile in LDS, and streams everything else from L2 into the
// registers the MFMAs consume:
//   * weights: a batch's offset is wave-uniform; the fragments of (offset, channel group) come from the STREAMED image
//     (prep_weights_kernel) 16 bytes per lane, once per run of batches with that offset;
//   * work unit ("stage") = up to two batches of one offset x one group of CG 16-channel input chunks
//     = 2 x CG x 4 x NBT MFMAs (128 at 64 -> 64: 4096 matrix-pipe cycles);
//   * software pipeline, one wave per SIMD: while stage s multiplies, the weights and rows of stage s+1 and the pair
//     entries of the chunk after that are in flight (the loads of a stage are issued one stage early, weights first, so
//     the MFMAs never wait for anything younger than what they need); the tile's chunk list (offset, first batch,
//     batch count) is derived once per tile from the entries and kept in LDS.
// Deterministic like the other kernels: batches in ascending offset order, one writer per accumulator tile.
constexpr int PSR = 128;  // rows per tile of the wide pair list
constexpr int PSW = 4;    // floats of padding per accumulator row: 16 rows x one 16-byte chunk spread over all 64 banks
constexpr int PSCH = 224; // chunk-list slots per wave (a 128-row tile has at most 27 * 8 = 216 batches)

__host__ __device__ constexpr size_t pairstream_wave_floats(int nbt) { return PSCH + (size_t)(PSR + 1) * (nbt * 16 + PSW); }

template <int NBT, int CG>
__global__ __launch_bounds__(512) void spconv_fwd_pairstream_kernel(ConvArgs p)
{
    struct { const float *bias, *bn_scale, *bn_shift, *bn_mean, *bn_invstd; } ep = {p.bias, p.bn_scale, p.bn_shift, p.bn_mean, p.bn_invstd};
    extern __shared__ float lds[];
    constexpr int CW = NBT * 16, RS = CW + PSW;  // accumulator row width / stride
    constexpr int F4 = CW / 4;                    // 16-byte chunks per row
    constexpr int RPS = 64 / F4;                  // rows the wave streams out per epilogue step
    const int l = lane_id(), q = l >> 4, jl = l & 15;
    const int waves = blockDim.x >> 6;
    const int nb0 = blockIdx.y * NBT;
    const int NG = p.NCH / CG;                    // channel groups (the launcher guarantees NCH % CG == 0)
    float *s_part = lds;                          // [2*Cout]
    int *s_next = reinterpret_cast<int *>(s_part + ((2 * p.Cout + 3) & ~3));
    float *s_pre = s_part + ((2 * p.Cout + 3) & ~3) + 4;   // [2][Cin]: scale, shift of the fused input BatchNorm
    float *wave_base = s_pre + ((2 * p.Cin + 3) & ~3) + (size_t)wave_id() * pairstream_wave_floats(NBT);
    int *s_chunks = reinterpret_cast<int *>(wave_base);    // (first batch) | (batches << 16) | (offset << 24)
    float *acc_t = wave_base + PSCH;                       // [(PSR + 1)][RS]

    const int nblk = gridDim.x;
    const int per_xcd = (nblk + 7) / 8;
    int vb = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;  // blocks of one XCD take neighbouring tiles
    if (nblk % 8 != 0) vb = blockIdx.x;
    const int *__restrict__ part_start = p.pl_tile_start + p.ntiles + 1;
    const int4 *__restrict__ picks = reinterpret_cast<const int4 *>(p.pl_tile_start + MS3D_PL_SCHED_OFFSET(p.ntiles));
    const int2 *__restrict__ entries = reinterpret_cast<const int2 *>(p.pl_entries);
    const f32x4 *__restrict__ wfs4 = reinterpret_cast<const f32x4 *>(p.wfs);
    const int t0 = part_start[vb], nmine = part_start[vb + 1] - t0;
    int slot = wave_id();
    int4 desc = picks[min(t0 + min(slot, max(nmine - 1, 0)), p.ntiles - 1)];
    const bool with_partial = p.bn_x != nullptr || p.out_stats != 0;
    if (threadIdx.x == 0) *s_next = waves;
    if (p.pre_scale)
        for (int c = threadIdx.x; c < p.Cin; c += blockDim.x) {
            s_pre[c] = p.pre_scale[c];
            s_pre[p.Cin + c] = p.pre_shift[c];
        }
    for (int e = l; e < (PSR + 1) * RS / 4; e += 64) reinterpret_cast<f32x4 *>(acc_t)[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    f32x4 st1 = {0.f, 0.f, 0.f, 0.f}, st2 = {0.f, 0.f, 0.f, 0.f};  // per-lane column sums (columns 4*(l % F4)..+3)
    const int c4 = l % F4, col = 16 * nb0 + 4 * c4;
    const bool ep_lane = l < RPS * F4;

    int round = 0;
    while (slot < nmine) {
        // fixed schedule (see spconv_fwd_pairlist_kernel): bit-reproducible statistics; MS3D_PL_DYNAMIC=1 = the pick counter
        ++round;
        int next_slot = round * waves + ((round & 1) ? waves - 1 - wave_id() : wave_id());
        if (p.dyn_picks) {
            int pick = 0;
            if (l == 0) pick = atomicAdd(s_next, 1);
            next_slot = pick;
        }
        const int tile = __builtin_amdgcn_readfirstlane(desc.x);
        const int row0 = tile * PSR;
        const int b_begin = __builtin_amdgcn_readfirstlane(desc.y);
        const int nbatch = min(__builtin_amdgcn_readfirstlane(desc.z) - b_begin, PSCH);
        const int2 *__restrict__ tile_entries = entries + (size_t)b_begin * 16;
        next_slot = __builtin_amdgcn_readfirstlane(next_slot);
        const int4 next_desc = picks[t0 + min(next_slot, nmine - 1)];

        // ---- chunk list of the tile: runs of batches with one offset, cut into chunks of <= 2 batches
        int nchunks = 0;
        {
            int kb[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int gi = l + 64 * j;
                kb[j] = gi < nbatch ? (tile_entries[(size_t)gi * 16].y >> 8) : 255;
            }
            int carry_k = -1, carry_rs = -1;   // offset of the batch before this register's first lane / its run start
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int gi = l + 64 * j;
                int prev = __shfl_up(kb[j], 1, 64);
                if (l == 0) prev = carry_k;
                int nxtk = __shfl_down(kb[j], 1, 64);
                const int first_next = j < 3 ? __builtin_amdgcn_readlane(kb[j < 3 ? j + 1 : 3], 0) : 255;
                if (l == 63) nxtk = first_next;
                const bool run_start = gi < nbatch && kb[j] != prev;
                int rs = run_start ? gi : -1;  // inclusive max-scan: start of the run this batch belongs to
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int o = __shfl_up(rs, d, 64);
                    if (l >= d) rs = max(rs, o);
                }
                rs = max(rs, carry_rs);
                const bool chunk_start = gi < nbatch && ((gi - rs) & 1) == 0;
                const int n = (gi + 1 < nbatch && nxtk == kb[j]) ? 2 : 1;
                const unsigned long long m = __ballot(chunk_start);
                if (chunk_start) s_chunks[nchunks + ballot_rank(m)] = gi | (n << 16) | (kb[j] << 24);
                nchunks += __popcll(m);
                carry_k = __builtin_amdgcn_readlane(kb[j], 63);
                carry_rs = __builtin_amdgcn_readlane(rs, 63);
            }
        }
        __builtin_amdgcn_wave_barrier();

        const int S = nchunks * NG;
        if (S > 0) {
            auto chunk_at = [&](int ci) { return __builtin_amdgcn_readfirstlane(s_chunks[min(ci, nchunks - 1)]); };
            auto load_entries = [&](int info) {
                const int c = info & 0xFFFF, n = (info >> 16) & 0xFF;
                return tile_entries[(size_t)(c + min((l >> 4) & 1, n - 1)) * 16 + jl];
            };
            auto load_w = [&](int k, int g, f32x4 (&w)[CG][NBT]) {
#pragma unroll
                for (int ch = 0; ch < CG; ch++)
#pragma unroll
                    for (int nb = 0; nb < NBT; nb++)
                        w[ch][nb] = wfs4[((size_t)(k * p.NCH + g * CG + ch) * p.NBtot + nb0 + nb) * 64 + l];
            };
            auto gather = [&](const int2 &e, int g, f32x4 (&a)[2][CG]) {
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    int in_row = __builtin_amdgcn_ds_bpermute((i * 16 + jl) << 2, e.x);
                    const float *row = p.in + (size_t)in_row * (size_t)p.Cin + 16 * g * CG + 4 * q;
#pragma unroll
                    for (int ch = 0; ch < CG; ch++) a[i][ch] = *reinterpret_cast<const f32x4 *>(row + 16 * ch);
                }
            };
            int info = chunk_at(0), info1 = chunk_at(1);
            int2 e_cur = load_entries(info), e_nxt = load_entries(info1), e_nxt2 = e_nxt;
            f32x4 w_cur[CG][NBT], w_nxt[CG][NBT], a_cur[2][CG], a_nxt[2][CG], d[2][NBT];
            load_w(info >> 24, 0, w_cur);
            gather(e_cur, 0, a_cur);
            int ci = 0, g = 0;
            for (int s = 0; s < S; s++) {
                const int n = (info >> 16) & 0xFF, k = info >> 24;
                const bool last_g = g == NG - 1;
                // ---- requests of the next stage: weights first, then rows, then the entries of the chunk after next
                const int g1 = last_g ? 0 : g + 1;
                const int k1 = last_g ? (info1 >> 24) : k;
                // (every stage issues the SAME loads -- the weights again even when the next chunk continues this offset,
                // the entries again for every channel group: a load behind a branch makes the wait in front of the MFMAs
                // cover the shorter path, i.e. wait for loads of THIS stage, and the pipeline is gone)
                load_w(k1, g1, w_nxt);
                gather(last_g ? e_nxt : e_cur, g1, a_nxt);
                e_nxt2 = load_entries(chunk_at(ci + 2));
                __builtin_amdgcn_sched_barrier(0);
                // ---- this stage
                if (g == 0) {
#pragma unroll
                    for (int i = 0; i < 2; i++)
#pragma unroll
                        for (int nb = 0; nb < NBT; nb++) d[i][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                if (p.pre_scale) {
#pragma unroll
                    for (int ch = 0; ch < CG; ch++) {
                        const int c0 = 16 * (g * CG + ch) + 4 * q;
                        const f32x4 sc = *reinterpret_cast<const f32x4 *>(s_pre + c0);
                        const f32x4 sh = *reinterpret_cast<const f32x4 *>(s_pre + p.Cin + c0);
#pragma unroll
                        for (int i = 0; i < 2; i++)
#pragma unroll
                            for (int t = 0; t < 4; t++) {
                                const float x = fmaf(a_cur[i][ch][t], sc[t], sh[t]);
                                a_cur[i][ch][t] = p.pre_relu ? fmaxf(x, 0.f) : x;
                            }
                    }
                }
#pragma unroll
                for (int ch = 0; ch < CG; ch++)
#pragma unroll
                    for (int t = 0; t < 4; t++)
#pragma unroll
                        for (int nb = 0; nb < NBT; nb++)
                            d[0][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w_cur[ch][nb][t], a_cur[0][ch][t], d[0][nb], 0, 0, 0);
                if (n > 1) {
#pragma unroll
                    for (int ch = 0; ch < CG; ch++)
#pragma unroll
                        for (int t = 0; t < 4; t++)
#pragma unroll
                            for (int nb = 0; nb < NBT; nb++)
                                d[1][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w_cur[ch][nb][t], a_cur[1][ch][t], d[1][nb], 0, 0, 0);
                }
                if (last_g) {
                    // D^T layout: row = output channel 4q + r, column = pair jl -> 16 contiguous bytes of the pair's row;
                    // plain read-modify-write (one writer per tile, LDS operations execute in order, pads -> dummy row)
#pragma unroll
                    for (int i = 0; i < 2; i++) {
                        if (i >= n) break;
                        const int orow = __builtin_amdgcn_ds_bpermute((i * 16 + jl) << 2, e_cur.y) & 255;
#pragma unroll
                        for (int nb = 0; nb < NBT; nb++) {
                            f32x4 *dst = reinterpret_cast<f32x4 *>(acc_t + orow * RS + 16 * nb + 4 * q);
                            f32x4 c = *dst;
#pragma unroll
                            for (int r = 0; r < 4; r++) c[r] += d[i][nb][r];
                            *dst = c;
                        }
                    }
                }
                // ---- rotate the pipeline registers
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int ch = 0; ch < CG; ch++) a_cur[i][ch] = a_nxt[i][ch];
#pragma unroll
                for (int ch = 0; ch < CG; ch++)
#pragma unroll
                    for (int nb = 0; nb < NBT; nb++) w_cur[ch][nb] = w_nxt[ch][nb];
                if (last_g) {
                    ci++;
                    g = 0;
                    info = info1;
                    info1 = chunk_at(ci + 1);
                    e_cur = e_nxt;
                    e_nxt = e_nxt2;
                } else {
                    g++;
                }
            }
        }
        slot = next_slot;
        desc = next_desc;
        __builtin_amdgcn_wave_barrier();
        // ---- epilogue: PSR x CW accumulator tile, row-major, 16 B per lane per step
        // (per-tile view of the per-column operands: hoisted out of the tile loop they sat in scratch across the pair loop)
        asm volatile("" : "+s"(ep.bias), "+s"(ep.bn_scale), "+s"(ep.bn_shift), "+s"(ep.bn_mean), "+s"(ep.bn_invstd));
        for (int r0 = 0; r0 < PSR; r0 += RPS) {
            const int r = r0 + l / F4;
            if (!ep_lane || r >= PSR) continue;
            f32x4 *src = reinterpret_cast<f32x4 *>(acc_t + r * RS + 4 * c4);
            f32x4 o4 = *src;
            *src = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int row = row0 + r;
            if (row < p.Vout) {
                const size_t o = (size_t)row * p.Cout + col;
                if (p.residual) {
                    const f32x4 rs = *reinterpret_cast<const f32x4 *>(p.residual + o);
#pragma unroll
                    for (int t = 0; t < 4; t++) o4[t] += rs[t];
                }
                if (ep.bias) {
#pragma unroll
                    for (int t = 0; t < 4; t++) o4[t] += ep.bias[col + t];
                }
                if (p.bn_x) {
                    const f32x4 x = *reinterpret_cast<const f32x4 *>(p.bn_x + o);
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        const float z = fmaf(x[t], ep.bn_scale[col + t], ep.bn_shift[col + t]);
                        const float gz = (z > 0.f) ? o4[t] : 0.f;
                        o4[t] = gz;
                        st1[t] += gz;
                        st2[t] += gz * ((x[t] - ep.bn_mean[col + t]) * ep.bn_invstd[col + t]);
                    }
                } else if (p.out_stats) {
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        st1[t] += o4[t];
                        st2[t] = fmaf(o4[t], o4[t], st2[t]);
                    }
                }
                *reinterpret_cast<f32x4 *>(p.out + o) = o4;
            }
        }
        // the dummy row collected the pad products; clear it with the rest
        if (l < F4) reinterpret_cast<f32x4 *>(acc_t + PSR * RS)[l] = (f32x4){0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_wave_barrier();
    }
    if (with_partial) {
        // lanes with equal l % F4 hold the same 4 columns (F4 need not be a power of two): through the wave's idle
        // accumulator tile, lane-major, then F4 lanes add the RPS copies up in a fixed order
        float *scr = acc_t;                       // [64][8] per-lane sums, then [2][CW] wave sums behind them
        *reinterpret_cast<f32x4 *>(scr + 8 * l) = ep_lane ? st1 : (f32x4){0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4 *>(scr + 8 * l + 4) = ep_lane ? st2 : (f32x4){0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_wave_barrier();
        if (l < F4) {
            f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = a1;
            for (int j = 0; j < RPS; j++) {
                const f32x4 v1 = *reinterpret_cast<const f32x4 *>(scr + 8 * (l + j * F4));
                const f32x4 v2 = *reinterpret_cast<const f32x4 *>(scr + 8 * (l + j * F4) + 4);
#pragma unroll
                for (int t = 0; t < 4; t++) { a1[t] += v1[t]; a2[t] += v2[t]; }
            }
            *reinterpret_cast<f32x4 *>(scr + 512 + 4 * l) = a1;
            *reinterpret_cast<f32x4 *>(scr + 512 + CW + 4 * l) = a2;
        }
        __syncthreads();
        float *dst = p.bn_partial + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 * p.Cout;
        const float *wave0 = s_pre + ((2 * p.Cin + 3) & ~3) + PSCH + 512;
        for (int t = threadIdx.x; t < 2 * p.Cout; t += blockDim.x) {
            const int which = t >= p.Cout, c = t - which * p.Cout - 16 * nb0;  // column inside this block's slice
            float sum = 0.f;
            if (c >= 0 && c < CW)
                for (int wv = 0; wv < waves; wv++) sum += wave0[(size_t)wv * pairstream_wave_floats(NBT) + which * CW + c];
            dst[t] = sum;
        }
    }
}

template <int NBT, int CG>
int launch_fwd_pairstream(ConvArgs p, dim3 grid, int threads, size_t lds, hipStream_t stream)
{
    p.ntiles = ms3d_divup(p.Vout, PSR);
    static const hipError_t attr = raise_lds_ceiling((const void *)spconv_fwd_pairstream_kernel<NBT, CG>);
    MS3D_CHECK(attr);
    spconv_fwd_pairstream_kernel<NBT, CG><<<grid, threads, lds, stream>>>(p);
    MS3D_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ backward-weight
struct WgradArgs {
    const float *in;         // [Vin, Cin]
    const float *dout;       // [Vout, Cout]
    const int *nbr;          // [K][Vout]
    float *partial;          // [gridDim.x][K, Cin, Cout] per-row-chunk partial sums (reduced by a second kernel)
    const float *pre_scale;  // fused BN(+ReLU) on `in` (recomputed), or null
    const float *pre_shift;
    int Vout, K, Cin, Cout, NBtot, rows_per_block, pre_relu;
    const int *ol_kt_start;  // offset-major pair list of the table (ms3d_kmap_offsetlist_build) or null
    const int *ol_entries;
};

// grid: x = row chunk, y = offset group (KG offsets), z = 16-channel input chunk.  Rows are the MFMA reduction
// dimension (4 per instruction).  All KG neighbour indices, then all KG gathers, are issued before the first
// MFMA so a wave keeps ~KG loads in flight; the KG offsets share the dout fragment.
template <int KG, int NBT>
__device__ __forceinline__ void wgrad_body(const WgradArgs &p, const int bx, const int by, const int bz, float *s_red)
{
    const int l = lane_id(), q = l >> 4, cl = l & 15;
    const int k0 = by * KG;
    const int c = bz * 16 + cl;  // input channel owned by this lane's A element
    const bool c_ok = c < p.Cin;
    const int c_safe = c_ok ? c : 0;
    const int c_mask = c_ok ? -1 : 0;
    const float sc = (p.pre_scale && c_ok) ? p.pre_scale[c] : 1.f;
    const float sh = (p.pre_scale && c_ok) ? p.pre_shift[c] : 0.f;
    f32x4 acc[KG][NBT];
#pragma unroll
    for (int a = 0; a < KG; a++)
#pragma unroll
        for (int b = 0; b < NBT; b++) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int r_begin = bx * p.rows_per_block;
    const int r_end = min(p.Vout, r_begin + p.rows_per_block);
    const int nw = blockDim.x >> 6;
    // Two independent 4-row steps per trip (rows r0.. and r0 + 4*nw..): all loads of both steps are issued before
    // the first MFMA, so a trip costs one dependent round trip (index -> row) for 8 rows instead of two.
    for (int r0 = r_begin + wave_id() * 4; r0 < r_end; r0 += 8 * nw) {
        int idx[2][KG];
        float a[2][KG], b[2][NBT];
        bool ok[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int row = r0 + h * 4 * nw + q;  // k-slot q of the MFMA <-> row
            ok[h] = row < r_end;
            const int safe_row = ok[h] ? row : r_begin;
#pragma unroll
            for (int nb = 0; nb < NBT; nb++) {
                const int j = 16 * nb + cl;
                const float v = p.dout[(size_t)safe_row * p.Cout + (j < p.Cout ? j : 0)];
                b[h][nb] = __int_as_float(__float_as_int(v) & ((ok[h] && j < p.Cout) ? -1 : 0));
            }
            // unconditional loads (clamped addresses, values masked afterwards): all gathers in flight together
#pragma unroll
            for (int kk = 0; kk < KG; kk++) {
                const int v = p.nbr[(size_t)min(k0 + kk, p.K - 1) * p.Vout + safe_row];
                idx[h][kk] = v | ((k0 + kk < p.K && ok[h]) ? 0 : -1);
            }
        }
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int kk = 0; kk < KG; kk++) a[h][kk] = p.in[(size_t)max(idx[h][kk], 0) * p.Cin + c_safe];
#pragma unroll
        for (int h = 0; h < 2; h++) {
#pragma unroll
            for (int kk = 0; kk < KG; kk++) {
                float v = a[h][kk];
                if (p.pre_scale) {
                    v = fmaf(v, sc, sh);
                    if (p.pre_relu) v = fmaxf(v, 0.f);
                }
                a[h][kk] = __int_as_float(__float_as_int(v) & ~(idx[h][kk] >> 31) & c_mask);
            }
#pragma unroll
            for (int kk = 0; kk < KG; kk++) {
                if (__ballot(idx[h][kk] >= 0) == 0ull) continue;
#pragma unroll
                for (int nb = 0; nb < NBT; nb++)
                    acc[kk][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[h][kk], b[h][nb], acc[kk][nb], 0, 0, 0);
            }
        }
    }
    // D layout: row (= input channel within the chunk) = 4q + reg, col (= output column) = cl.
    // Cross-wave sum through LDS, one offset (NBT accumulators) per barrier pair, all threads take part in the sum;
    // fixed wave order -> deterministic.  One plain store per element per block.
    float *dst = p.partial + (size_t)bx * p.K * p.Cin * p.Cout;
#pragma unroll
    for (int kk = 0; kk < KG; kk++) {
        const int k = k0 + kk;
        __syncthreads();
#pragma unroll
        for (int nb = 0; nb < NBT; nb++)
#pragma unroll
            for (int r = 0; r < 4; r++) s_red[(wave_id() * NBT + nb) * 256 + r * 64 + l] = acc[kk][nb][r];
        __syncthreads();
        for (int e = threadIdx.x; e < NBT * 256; e += blockDim.x) {
            float v = 0.f;
            for (int w = 0; w < nw; w++) v += s_red[w * NBT * 256 + e];
            const int nb = e >> 8, r = (e >> 6) & 3, ln = e & 63;
            const int ci = bz * 16 + 4 * (ln >> 4) + r, j = 16 * nb + (ln & 15);
            if (k < p.K && ci < p.Cin && j < p.Cout) dst[((size_t)k * p.Cin + ci) * p.Cout + j] = v;
        }
    }
}

template <int KG, int NBT>
__global__ __launch_bounds__(256) void spconv_wgrad_kernel(WgradArgs p)
{
    __shared__ float s_red[4 * NBT * 256];  // every wave parks the NBT accumulators of one offset (4 regs x 64 lanes each)
    wgrad_body<KG, NBT>(p, blockIdx.x, blockIdx.y, blockIdx.z, s_red);
}

// The same kernel for MANY layers in one launch (ms3d_spconv_wgrad_multi): the backward-weight of a layer needs its
// input and its output gradient only, nothing downstream waits for it but the optimizer -- so the launches of the small
// levels at the bottom of the U-Net (a few hundred to a few thousand rows: 25-40 us each, most of it launch floor, with
// a fraction of the chip busy) are queued during the backward pass and run together, all layers of one (KG, NBT) shape
// class side by side.  A block finds its layer in the descriptor table (blocks are numbered layer after layer).
struct WgradLaunch {
    int block_begin;   // first block of this layer in the batched launch (patched by the host when the table is built)
    int gx, gy, gz;    // the layer's own grid
    int variant;       // KG * 100 + NBT
    int nblk;          // slabs it leaves in p.partial
    int pad[2];
    WgradArgs p;
};
static_assert(sizeof(WgradLaunch) <= 128, "ms3d_spconv_wgrad_launch_bytes() promises 128");

template <int KG, int NBT>
__global__ __launch_bounds__(256) void spconv_wgrad_multi_kernel(const unsigned char *__restrict__ descs, int n_desc)
{
    __shared__ float s_red[4 * NBT * 256];
    int lo = 0, hi = n_desc - 1;
    const int blk = blockIdx.x;
    while (lo < hi) {        // last descriptor whose first block is <= blk
        const int mid = (lo + hi + 1) >> 1;
        if (reinterpret_cast<const WgradLaunch *>(descs + 128 * (size_t)mid)->block_begin <= blk) lo = mid; else hi = mid - 1;
    }
    const WgradLaunch d = *reinterpret_cast<const WgradLaunch *>(descs + 128 * (size_t)lo);
    const int local = blk - d.block_begin;
    wgrad_body<KG, NBT>(d.p, local % d.gx, (local / d.gx) % d.gy, local / (d.gx * d.gy), s_red);
}

// ---- backward-weight on three-piece bf16 operands (wide submanifold layers) --------------------------------------
// The reduction dimension of dW[k] = sum_rows act(in[nbr_k(row)])^T dout[row] is the ROW: v_mfma_f32_16x16x32_bf16 takes 32
// rows per instruction (the f32 instruction 4) and wants, per lane, EIGHT consecutive rows of its channel -- a column of
// the row-major tensors.  Two elementwise passes per layer prepare both operands (exact three-way split, see write_bf3):
//   dout -> its MFMA operands ready made (it is not gathered): img[((t*NB + nb)*3 + piece)*64 + lane][e] =
//           piece of dout[32 t + 8 (lane >> 4) + e][16 nb + (lane & 15)]                  (split_rows_bf3)
//   x    -> the ACTIVATED input act(x) = relu(x * scale + shift) as pieces, row-major: xs[(row*3 + piece)*C + c]
//           (act_split_bf3; one launch does both: wgrad_bf3_operands_kernel)
// A workgroup owns (row chunk, KG offsets, one 16-channel input chunk) and walks its rows 32 at a time.  Its waves SPLIT THE
// OUTPUT COLUMNS (NBW blocks of 16 each) and share the gathered input: per trip the whole workgroup gathers the KG tiles of
// 32 rows x 16 channels x 3 pieces (one 16-byte word per thread and entry, no word fetched twice), parks them ROW-major in
// LDS, and every wave reads its MFMA operand back with the transposing LDS read: ds_read_b64_tr_b16 hands lane i of a
// 16-lane group the element (i & 3) of the 8-byte chunk that lane 4r + (i >> 2) of the group addressed, r = 0..3 -- with
// lane s addressing row base + (s >> 2), channels 4 (s & 3) .. +3, lane i receives rows base .. base + 3 of channel i
// (tools/probe/tr_probe.hip prints the mapping).  A wave's dout operands (its own columns only) stay in registers for the
// trip; the next trip's gathers, indices and dout operands are in flight during this trip's MFMAs.  Six earlier builds
// (every wave gathering and transposing for itself; dout operands re-read from LDS per offset) were slower than the f32
// kernel on operand traffic alone (profiles/r03_fwd_experiments.txt section 6).  No cross-wave reduction: every
// accumulator belongs to one (offset, column block).  rows_per_block is a multiple of 32.
__device__ __forceinline__ void split_rows_bf3(long blk, const float *__restrict__ x, long V, int C, int NB, long ntile,
                                               bf16x8 *__restrict__ img)
{
    const long o = blk * 256 + threadIdx.x;          // (t, nb, lane)
    if (o >= ntile * NB * 64) return;
    const int lane = (int)(o & 63);
    const long tn = o >> 6;
    const int nb = (int)(tn % NB);
    const long t = tn / NB;
    const int j = 16 * nb + (lane & 15);
    bf16x8 p0, p1, p2;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const long row = 32 * t + 8 * (lane >> 4) + e;
        const float v = (row < V && j < C) ? x[row * C + j] : 0.f;
        __bf16 h0, h1, h2;
        split3(v, h0, h1, h2);
        p0[e] = h0; p1[e] = h1; p2[e] = h2;
    }
    bf16x8 *dst = img + (tn * 3) * 64 + lane;
    dst[0] = p0; dst[64] = p1; dst[128] = p2;
}

__device__ __forceinline__ void act_split_bf3(long blk, const float *__restrict__ x, long V, int C,
                                              const float *__restrict__ scale, const float *__restrict__ shift, int relu,
                                              bf16x8 *__restrict__ xs)
{
    const int C8 = C >> 3;
    const long o = blk * 256 + threadIdx.x;          // (row, 8-channel group)
    if (o >= V * C8) return;
    const long row = o / C8;
    const int c0 = (int)(o % C8) * 8;
    const f32x4 a = *reinterpret_cast<const f32x4 *>(x + row * C + c0), b = *reinterpret_cast<const f32x4 *>(x + row * C + c0 + 4);
    float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    bf16x8 p0, p1, p2;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        float y = v[e];
        if (scale) {
            y = fmaf(y, scale[c0 + e], shift[c0 + e]);
            if (relu) y = fmaxf(y, 0.f);
        }
        __bf16 h0, h1, h2;
        split3(y, h0, h1, h2);
        p0[e] = h0; p1[e] = h1; p2[e] = h2;
    }
    bf16x8 *dst = xs + (row * 3) * C8 + (c0 >> 3);
    dst[0] = p0; dst[C8] = p1; dst[2 * C8] = p2;
}

// both passes in one launch: blocks [0, nblk_dout) lay dout out, the rest split the activated input
__global__ __launch_bounds__(256) void wgrad_bf3_operands_kernel(const float *__restrict__ dout, const float *__restrict__ x, long V,
                                                                 int Cin, int Cout, int NB, long ntile, int nblk_dout,
                                                                 const float *__restrict__ scale, const float *__restrict__ shift,
                                                                 int relu, bf16x8 *__restrict__ img, bf16x8 *__restrict__ xs)
{
    if ((int)blockIdx.x < nblk_dout) split_rows_bf3(blockIdx.x, dout, V, Cout, NB, ntile, img);
    else act_split_bf3((long)blockIdx.x - nblk_dout, x, V, Cin, scale, shift, relu, xs);
}

struct WgradBf3Args {
    const bf16x8 *xs;        // activated input pieces, row-major [Vin][3][Cin/8]
    const bf16x8 *dout_img;  // dout operands [Vout/32][NB][3][64]
    const int *nbr;          // [K][Vout]
    float *partial;
    int Vout, K, Cin, Cout, NBtot, rows_per_block;
};

typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int KG, int NW, int NBW>
__global__ __launch_bounds__(64 * NW, 2) void spconv_wgrad_bf3_kernel(WgradBf3Args p)
{
    constexpr int NT = 64 * NW, TILE = 192;                  // a tile = 32 rows x (3 pieces x 2 halves) 16-byte words
    constexpr int E = (KG * TILE + NT - 1) / NT;             // gather entries per thread and trip
    __shared__ __attribute__((aligned(16))) short s_a[KG][3][32 * 16];          // [offset][piece][row][16 ch]
    const int l = lane_id(), q = l >> 4, cl = l & 15, w = wave_id();
    const int k0 = blockIdx.y * KG;
    const int C8 = p.Cin >> 3;
    f32x4 acc[KG][NBW];
#pragma unroll
    for (int a = 0; a < KG; a++)
#pragma unroll
        for (int b = 0; b < NBW; b++) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int r_begin = blockIdx.x * p.rows_per_block;
    const int r_end = min(p.Vout, r_begin + p.rows_per_block);
    const int rd0 = (8 * q + (cl >> 2)) * 16 + 4 * (cl & 3), rd1 = rd0 + 4 * 16;
    // entry e of the workgroup's gather list: offset e / 192, row (e % 192) / 6, piece ((e % 6) >> 1), half e & 1 -- packed
    // into one register per entry: bits 0-14 LDS slot (in 16-byte words... shorts / 8), 15-19 row, 20-27 source word, 28-32 offset
    unsigned e_pk[E];
#pragma unroll
    for (int i = 0; i < E; i++) {
        const int e = threadIdx.x + i * NT;
        const int kk = min(e / TILE, KG - 1), rem = e % TILE, row = rem / 6, pc = (rem % 6) >> 1, half = rem & 1;
        const bool ok = e < KG * TILE && k0 + kk < p.K;
        e_pk[i] = (unsigned)(((kk * 3 + pc) * 512 + row * 16 + half * 8) >> 3) | ((unsigned)row << 15)
                  | ((unsigned)(pc * C8 + blockIdx.z * 2 + half) << 20) | ((unsigned)(ok ? kk : 15) << 28);
    }
    auto load_idx = [&](int r0, int (&idx)[E]) {
#pragma unroll
        for (int i = 0; i < E; i++) {
            const int row = r0 + (int)((e_pk[i] >> 15) & 31), kk = (int)(e_pk[i] >> 28);
            const bool ok = kk != 15 && row < r_end;
            const int v = p.nbr[(size_t)(k0 + (ok ? kk : 0)) * p.Vout + (ok ? row : r_begin)];
            idx[i] = v | (ok ? 0 : -1);
        }
    };
    auto gather = [&](const int (&idx)[E], uint4 (&g)[E]) {
#pragma unroll
        for (int i = 0; i < E; i++) {
            const uint4 v = reinterpret_cast<const uint4 *>(p.xs)[(size_t)max(idx[i], 0) * 3 * C8 + ((e_pk[i] >> 20) & 255)];
            const unsigned keep = idx[i] >= 0 ? 0xffffffffu : 0u;
            g[i] = make_uint4(v.x & keep, v.y & keep, v.z & keep, v.w & keep);
        }
    };
    auto load_b = [&](int r0, uint4 (&bw)[NBW][3]) {
#pragma unroll
        for (int j = 0; j < NBW; j++) {
            const int nb = w * NBW + j;
            const uint4 *src = reinterpret_cast<const uint4 *>(p.dout_img) + (((size_t)(r0 >> 5) * p.NBtot + min(nb, p.NBtot - 1)) * 3) * 64 + l;
            const bool ok = nb < p.NBtot && r0 < r_end;
#pragma unroll
            for (int pc = 0; pc < 3; pc++) bw[j][pc] = ok ? src[pc * 64] : make_uint4(0, 0, 0, 0);
        }
    };
    int idx_n[E];
    uint4 g[E];
    uint4 bw[NBW][3];
    {
        int idx_c[E];
        load_idx(r_begin, idx_c);
        gather(idx_c, g);
    }
    load_idx(r_begin + 32, idx_n);
    load_b(r_begin, bw);
    short *tiles = &s_a[0][0][0];
    for (int r0 = r_begin; r0 < r_end; r0 += 32) {
        __syncthreads();                         // everybody has read the previous trip's tiles
#pragma unroll
        for (int i = 0; i < E; i++)
            if (E * NT == KG * TILE || threadIdx.x + i * NT < KG * TILE) *reinterpret_cast<uint4 *>(tiles + (e_pk[i] & 0x7fff) * 8) = g[i];
        __syncthreads();
        uint4 bc[NBW][3];
#pragma unroll
        for (int j = 0; j < NBW; j++)
#pragma unroll
            for (int pc = 0; pc < 3; pc++) bc[j][pc] = bw[j][pc];
        // the next trip's operands fly during this trip's MFMAs
        gather(idx_n, g);
        load_idx(r0 + 64, idx_n);
        load_b(r0 + 32, bw);
#pragma unroll
        for (int kk = 0; kk < KG; kk++) {
            bf16x8 a[3];
#pragma unroll
            for (int pc = 0; pc < 3; pc++) {
                const short *tp = tiles + (kk * 3 + pc) * 512;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(tp + rd0));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(tp + rd1));
                short t8[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                a[pc] = *reinterpret_cast<const bf16x8 *>(t8);
            }
#pragma unroll
            for (int j = 0; j < NBW; j++) {
                const bf16x8 w0 = *reinterpret_cast<const bf16x8 *>(&bc[j][0]), w1 = *reinterpret_cast<const bf16x8 *>(&bc[j][1]),
                             w2 = *reinterpret_cast<const bf16x8 *>(&bc[j][2]);
                acc[kk][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], w0, acc[kk][j], 0, 0, 0);
                acc[kk][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], w2, acc[kk][j], 0, 0, 0);
                acc[kk][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], w1, acc[kk][j], 0, 0, 0);
                acc[kk][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], w0, acc[kk][j], 0, 0, 0);
                acc[kk][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], w1, acc[kk][j], 0, 0, 0);
                acc[kk][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], w0, acc[kk][j], 0, 0, 0);
            }
        }
    }
    // D layout: row (= input channel within the chunk) = 4q + reg, col (= output column) = cl
    float *dst = p.partial + (size_t)blockIdx.x * p.K * p.Cin * p.Cout;
#pragma unroll
    for (int kk = 0; kk < KG; kk++) {
        const int k = k0 + kk;
        if (k >= p.K) continue;
#pragma unroll
        for (int j = 0; j < NBW; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int ci = blockIdx.z * 16 + 4 * q + r, col = 16 * (w * NBW + j) + cl;
                if (ci < p.Cin && col < p.Cout) dst[((size_t)k * p.Cin + ci) * p.Cout + col] = acc[kk][j][r];
            }
    }
}

template <int KG, int NW, int NBW>
int launch_wgrad_bf3(const WgradBf3Args &p, int nblk_rows, hipStream_t stream)
{
    dim3 grid(nblk_rows, ms3d_divup(p.K, KG), ms3d_divup(p.Cin, 16));
    spconv_wgrad_bf3_kernel<KG, NW, NBW><<<grid, 64 * NW, 0, stream>>>(p);
    MS3D_LAUNCH_CHECK();
    return 0;
}

// dW[e] = sum over row chunks of partial[b][e].  Block = 16 elements x 16 slab lanes: lane j sums slabs j, j+16, ...
// and the 16 lane sums are combined in lane order -> fixed order, deterministic; 16x the threads of one-thread-per-
// element (the slab walk is latency-bound: 251 slabs took 21 us).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ partial, int nblk, long n,
                                                           float *__restrict__ dW)
{
    __shared__ float s_sum[16][17];
    const int el = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const long e = (long)blockIdx.x * 16 + el;
    float s0 = 0.f, s1 = 0.f;
    if (e < n) {
        int b = sl;
        for (; b + 16 < nblk; b += 32) {
            s0 += partial[(size_t)b * n + e];
            s1 += partial[(size_t)(b + 16) * n + e];
        }
        if (b < nblk) s0 += partial[(size_t)b * n + e];
    }
    s_sum[sl][el] = s0 + s1;
    __syncthreads();
    if (threadIdx.x < 16 && e < n) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 16; j++) t += s_sum[j][threadIdx.x];
        dW[e] = t;
    }
}

// The same for large images (wide layers: n = K * Cin * Cout >= 32k floats): 16 bytes per lane.  A block = 64 groups of
// 4 consecutive elements x 4 slab lanes; a slab lane reads 256 contiguous bytes per wave row and walks every 4th slab
// with 4 loads in flight; the 4 slab lanes are combined in lane order -> fixed order, deterministic.  The 16-element
// kernel above reads 64-byte pieces (17-21 us for 128 slabs of 442 KB; this one ~10 us).
__global__ __launch_bounds__(256) void wgrad_reduce4_kernel(const float *__restrict__ partial, int nblk, long n4,
                                                            float *__restrict__ dW)
{
    const int g = threadIdx.x & 15, sl = (threadIdx.x >> 4) & 3, sub = threadIdx.x >> 6;   // 16 groups x 4 slab lanes per wave
    const long e4 = (long)blockIdx.x * 64 + sub * 16 + g;                                  // float4 index
    const float4 *p4 = reinterpret_cast<const float4 *>(partial);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e4 < n4) {
        int b = sl;
        for (; b + 12 < nblk; b += 16) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = p4[(size_t)(b + 4 * u) * n4 + e4];
#pragma unroll
            for (int u = 0; u < 4; u++) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        for (; b < nblk; b += 4) {
            const float4 v = p4[(size_t)b * n4 + e4];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    // slab lanes 0..3 of a group sit 16 lanes apart: fold in fixed order (0 + 1) + (2 + 3)
#pragma unroll
    for (int d = 16; d <= 32; d <<= 1) {
        s.x += __shfl_xor(s.x, d, 64); s.y += __shfl_xor(s.y, d, 64);
        s.z += __shfl_xor(s.z, d, 64); s.w += __shfl_xor(s.w, d, 64);
    }
    if (sl == 0 && e4 < n4) reinterpret_cast<float4 *>(dW)[e4] = s;
}

static inline void launch_wgrad_reduce(const float *partial, int nblk, long n, float *dW, hipStream_t stream)
{
    if (n >= 32768 && (n & 3) == 0 && ((uintptr_t)partial & 15) == 0 && ((uintptr_t)dW & 15) == 0)
        wgrad_reduce4_kernel<<<ms3d_divup(n / 4, 64), 256, 0, stream>>>(partial, nblk, n / 4, dW);
    else
        wgrad_reduce_kernel<<<ms3d_divup(n, 16), 256, 0, stream>>>(partial, nblk, n, dW);
}

// ONE launch for the slab reductions of many layers (ms3d_wgrad_reduce_multi): a block finds its layer in the
// descriptor table (blocks are numbered layer after layer) and does exactly what a block of wgrad_reduce_kernel /
// wgrad_reduce4_kernel does for that layer -- same lanes, same summation order, bit-identical dW.
struct WgradReduceDesc {
    const float *slabs;
    float *dW;
    long n;           // floats per slab
    int nblk;         // slabs
    int block_begin;  // first block of this layer; bit 31 set: the 16-byte kernel's geometry (64 float4 groups per block)
};
static_assert(sizeof(WgradReduceDesc) == 32, "descriptor layout is part of the C ABI (host code packs it)");

__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(const WgradReduceDesc *__restrict__ descs, int n_desc)
{
    __shared__ float s_sum[16][17];
    int lo = 0, hi = n_desc - 1;
    const int blk = blockIdx.x;
    while (lo < hi) {        // last descriptor whose first block is <= blk
        const int mid = (lo + hi + 1) >> 1;
        if ((descs[mid].block_begin & 0x7fffffff) <= blk) lo = mid; else hi = mid - 1;
    }
    const WgradReduceDesc d = descs[lo];
    const int b0 = blk - (d.block_begin & 0x7fffffff);
    const float *__restrict__ partial = d.slabs;
    const int nblk = d.nblk;
    if (d.block_begin < 0) {
        const long n4 = d.n >> 2;
        const int g = threadIdx.x & 15, sl = (threadIdx.x >> 4) & 3, sub = threadIdx.x >> 6;
        const long e4 = (long)b0 * 64 + sub * 16 + g;
        const float4 *p4 = reinterpret_cast<const float4 *>(partial);
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e4 < n4) {
            int b = sl;
            for (; b + 12 < nblk; b += 16) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) v[u] = p4[(size_t)(b + 4 * u) * n4 + e4];
#pragma unroll
                for (int u = 0; u < 4; u++) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
            }
            for (; b < nblk; b += 4) {
                const float4 v = p4[(size_t)b * n4 + e4];
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        }
#pragma unroll
        for (int dd = 16; dd <= 32; dd <<= 1) {
            s.x += __shfl_xor(s.x, dd, 64); s.y += __shfl_xor(s.y, dd, 64);
            s.z += __shfl_xor(s.z, dd, 64); s.w += __shfl_xor(s.w, dd, 64);
        }
        if (sl == 0 && e4 < n4) reinterpret_cast<float4 *>(d.dW)[e4] = s;
        return;
    }
    const int el = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const long n = d.n, e = (long)b0 * 16 + el;
    float s0 = 0.f, s1 = 0.f;
    if (e < n) {
        int b = sl;
        for (; b + 16 < nblk; b += 32) {
            s0 += partial[(size_t)b * n + e];
            s1 += partial[(size_t)(b + 16) * n + e];
        }
        if (b < nblk) s0 += partial[(size_t)b * n + e];
    }
    s_sum[sl][el] = s0 + s1;
    __syncthreads();
    if (threadIdx.x < 16 && e < n) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 16; j++) t += s_sum[j][threadIdx.x];
        d.dW[e] = t;
    }
}

template <int KG, int NBT>
int launch_wgrad(const WgradArgs &p, int nblk_rows, hipStream_t stream, WgradLaunch *defer = nullptr)
{
    dim3 grid(nblk_rows, ms3d_divup(p.K, KG), ms3d_divup(p.Cin, 16));
    if (defer) {     // not launched: described for a later batched launch
        defer->block_begin = 0;
        defer->gx = grid.x; defer->gy = grid.y; defer->gz = grid.z;
        defer->variant = KG * 100 + NBT;
        defer->nblk = nblk_rows;
        defer->p = p;
        return 0;
    }
    spconv_wgrad_kernel<KG, NBT><<<grid, 256, 0, stream>>>(p);
    MS3D_LAUNCH_CHECK();
    return 0;
}

template <int KG, int NBT>
int launch_wgrad_multi(const void *descs, int n_desc, int total_blocks, hipStream_t stream)
{
    spconv_wgrad_multi_kernel<KG, NBT><<<total_blocks, 256, 0, stream>>>(static_cast<const unsigned char *>(descs), n_desc);
    MS3D_LAUNCH_CHECK();
    return 0;
}

// Backward-weight over an offset-major pair list.  Every MFMA step multiplies 4 REAL pairs (the table walk above spends
// a step on 4 consecutive output rows of which ~1 in 6 has the neighbour at full resolution).
// Block = one chunk of output tiles x ALL offsets (x one 16-channel input chunk).  The chunk's pairs, concatenated in
// offset order, are cut into equal slices, one per wave.  A wave that owns all of an offset's pairs stores that
// offset's slab directly; offsets cut by a slice boundary park their partial sums in LDS and are combined in wave
// order after one barrier: deterministic.
// Row gathers: the MFMA wants A[m = ci][k = pair] / B[k = pair][n = co], i.e. ONE float per lane and step; as global
// loads that is 2 dword gathers per 4 pairs (measured 64 us for 2.3 M pairs).  Instead a lane fetches 16 B of its
// pair's row (16 pairs x 64 B per instruction, as in the forward kernel), the wave parks the 16x16 tiles in its
// private LDS area and reads them back transposed (ds_read_b32, conflict-free): 4x fewer vector-memory instructions
// (55 us).  Tried and measured slower: a grid over (offset, pair range) (63 us), waves that own fixed offsets and
// sweep the chunk in lock step with a barrier per 256 rows (L2 requests / 3, but 82 us: two dependent round trips
// per sub-chunk), 4 instead of 2 batches per trip (60 us).
constexpr int WGRAD_TB = 2;  // 16-pair batches per trip

// NCH = 16-channel input chunks handled by one workgroup (blockIdx.z selects the group): the entries and the dout rows
// of a batch are fetched once for all of them instead of once per chunk.
// waves a workgroup of this instantiation can hold (150 KB of LDS, <= 16): its launch bound, so that the variants with large
// per-wave areas get the registers of the waves they cannot run anyway
constexpr int wgrad_list_wave_kb(int nbt, int nch, int tb) { return tb * (nch + nbt) > 2 * nch * nbt ? tb * (nch + nbt) : 2 * nch * nbt; }
constexpr int wgrad_list_max_waves(int nbt, int nch, int tb) { return 150 / wgrad_list_wave_kb(nbt, nch, tb) > 16 ? 16 : 150 / wgrad_list_wave_kb(nbt, nch, tb); }

template <int NBT, int NCH, int TB = WGRAD_TB>
__global__ __launch_bounds__(64 * wgrad_list_max_waves(NBT, NCH, TB)) void spconv_wgrad_offsetlist_kernel(WgradArgs p)
{
    constexpr int NT = NCH + NBT;   // 16x16 tiles parked per batch
    constexpr int NA = NCH * NBT;   // accumulators
    constexpr int WAVE_KB = TB * NT > 2 * NA ? TB * NT : 2 * NA;   // the wave's LDS area: tiles, later its boundary partials
    extern __shared__ float lds[];
    __shared__ int s_lo[32], s_cum[33], s_pk[16][2];
    const int l = lane_id(), q = l >> 4, cl = l & 15;
    const int nw = blockDim.x >> 6, w = wave_id();
    // per wave: [TB][NCH + NBT][16 pairs][16] transposition tiles; once the wave's pair loop is over the same area takes its (at
    // most two) boundary partials [2][NA*256] -- 2 NA <= TB (NCH + NBT) for every instantiation.  (Rounds 1-5 kept a separate
    // [nw][2][NA*256] area: 8 of the 18 KB per wave at 64 -> 64, which left 8 waves per CU -- round 6 counters: 44 % of the wave
    // cycles waiting, MFMA pipe 36 % busy.)  The FIRST partial of a wave is done long before its loop ends: it waits in registers.
    float *s_tile = lds + (size_t)w * WAVE_KB * 256;
    const int tiles = (p.Vout + MS3D_PL_ROWS - 1) / MS3D_PL_ROWS;
    // tile range of this workgroup: 2^rows_per_block consecutive parts of the list's equal-pair-count cut
    const int *__restrict__ part_start = p.ol_kt_start + (size_t)p.K * tiles + 1;
    const int t_lo = part_start[blockIdx.x << p.rows_per_block], t_hi = part_start[(blockIdx.x + 1) << p.rows_per_block];
    if (w == 0) {
        const int kk = min(l, p.K - 1);
        const int lo = p.ol_kt_start[(size_t)kk * tiles + t_lo], hi = p.ol_kt_start[(size_t)kk * tiles + t_hi];
        const int n = l < p.K ? hi - lo : 0;
        const int incl = wave_incl_scan(n);
        if (l < p.K) {
            s_lo[l] = lo;
            s_cum[l] = incl - n;
        }
        if (l == p.K - 1) s_cum[p.K] = incl;
    }
    if (threadIdx.x < 32) s_pk[threadIdx.x >> 1][threadIdx.x & 1] = -1;
    __syncthreads();

    // Transposition tiles are [16 pairs][4 chunks of 16 B]: row `cl` keeps chunk q at slot q ^ ((cl >> 1) & 3).  Unswizzled,
    // the eight lanes of a ds_write_b128 group (same q, pairs 0-7 / 8-15) are 64 B apart and hit two 16-byte bank ranges of
    // the 32-bank store path: 4-way conflicts on every park (round 6 counters: SQ_LDS_BANK_CONFLICT 61 M cycles per
    // 64 -> 64 launch at 196k rows, a third of the LDS time); the transposed ds_read_b32 (rows 4st+q, two rows per 32-lane
    // group that share (row >> 1)) stays a permutation of 32 consecutive dwords.
    const int qs = q ^ ((cl >> 1) & 3);
    const int cbase = blockIdx.z * NCH * 16;  // first input channel of this workgroup
    const int col0 = blockIdx.y * NBT * 16;   // first output column (round 5: layers beyond 64 output channels in slices of NBT blocks)
    f32x4 sc[NCH], sh[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) {
        const int c0 = cbase + 16 * ch + 4 * q;
        const bool ok = p.pre_scale && c0 < p.Cin;
        sc[ch] = ok ? *reinterpret_cast<const f32x4 *>(p.pre_scale + c0) : (f32x4){1.f, 1.f, 1.f, 1.f};
        sh[ch] = ok ? *reinterpret_cast<const f32x4 *>(p.pre_shift + c0) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const int2 *__restrict__ entries = reinterpret_cast<const int2 *>(p.ol_entries);
    float *slab = p.partial + (size_t)blockIdx.x * p.K * p.Cin * p.Cout;
    int opaque0 = 0;
    asm volatile("" : "+s"(opaque0));
    // D layout: row (input channel within its chunk) = 4q + reg, column (output channel) = cl
    auto store_slab = [&](int k, const f32x4 (&acc)[NA]) {
#pragma unroll
        for (int ch = 0; ch < NCH; ch++)
#pragma unroll
            for (int nb = 0; nb < NBT; nb++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int ci = cbase + 16 * ch + 4 * q + r, j = col0 + 16 * nb + cl;
                    if (ci < p.Cin && j < p.Cout) slab[((size_t)k * p.Cin + ci) * p.Cout + j] = acc[ch * NBT + nb][r];
                }
    };

    const int T = s_cum[p.K];
    const int s0 = (int)((long long)T * w / nw), s1 = (int)((long long)T * (w + 1) / nw);
    int n_partial = 0;
    f32x4 part0[NA];          // the slice's first cut offset
    int pk0 = -1;
#pragma unroll
    for (int b = 0; b < NA; b++) part0[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < p.K; k++) {
        const int c_lo = s_cum[k], c_hi = s_cum[k + 1];
        const int g_lo = max(s0, c_lo), g_hi = min(s1, c_hi);
        if (g_lo >= g_hi) continue;  // wave-uniform
        const int pbase = s_lo[k] - c_lo;  // list position of concatenated index g = pbase + g
        const int p_begin = pbase + g_lo, p_end = pbase + g_hi;
        f32x4 acc[NA];
#pragma unroll
        for (int b = 0; b < NA; b++) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // pipeline: entries two trips ahead, row gathers one trip ahead of the MFMAs (`opaque0` keeps the compiler
        // from merging the in-loop loads with the ones in front of the loop, which would undo the prefetch)
        int2 e[TB];
        f32x4 ga[TB][NCH], gb[TB][NBT];
        auto load_entries = [&](int base) {
#pragma unroll
            for (int t = 0; t < TB; t++) e[t] = entries[min(base + 16 * t + cl, p_end - 1)];
        };
        auto gather = [&]() {
#pragma unroll
            for (int t = 0; t < TB; t++) {
#pragma unroll
                for (int ch = 0; ch < NCH; ch++)
                    ga[t][ch] = *reinterpret_cast<const f32x4 *>(p.in + (size_t)e[t].x * p.Cin + min(cbase + 16 * ch, p.Cin - 16) + 4 * q);
#pragma unroll
                for (int nb = 0; nb < NBT; nb++)
                    gb[t][nb] = *reinterpret_cast<const f32x4 *>(p.dout + (size_t)e[t].y * p.Cout + min(col0 + 16 * nb, p.Cout - 16) + 4 * q);
            }
        };
        load_entries(p_begin);
        gather();
        load_entries(p_begin + 16 * TB + opaque0);
        for (int base = p_begin; base < p_end; base += 16 * TB) {
            f32x4 ca[TB][NCH], cb[TB][NBT];
#pragma unroll
            for (int t = 0; t < TB; t++) {
#pragma unroll
                for (int ch = 0; ch < NCH; ch++) ca[t][ch] = ga[t][ch];
#pragma unroll
                for (int nb = 0; nb < NBT; nb++) cb[t][nb] = gb[t][nb];
            }
            gather();                                        // rows of trip + 1 (entries already here)
            load_entries(base + 2 * 16 * TB + opaque0);      // entries of trip + 2
            // current trip: activation + zero the pairs beyond the range, park, read back transposed, multiply
#pragma unroll
            for (int t = 0; t < TB; t++) {
                const bool ok = base + 16 * t + cl < p_end;
                float *ta = s_tile + (size_t)t * NT * 256;
#pragma unroll
                for (int ch = 0; ch < NCH; ch++) {
                    const bool ch_ok = ok && cbase + 16 * ch < p.Cin;
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        float v = ca[t][ch][i];
                        if (p.pre_scale) {
                            v = fmaf(v, sc[ch][i], sh[ch][i]);
                            if (p.pre_relu) v = fmaxf(v, 0.f);
                        }
                        ca[t][ch][i] = ch_ok ? v : 0.f;
                    }
                    *reinterpret_cast<f32x4 *>(ta + ch * 256 + cl * 16 + 4 * qs) = ca[t][ch];
                }
#pragma unroll
                for (int nb = 0; nb < NBT; nb++)
                    *reinterpret_cast<f32x4 *>(ta + (NCH + nb) * 256 + cl * 16 + 4 * qs) = cb[t][nb];
            }
#pragma unroll
            for (int t = 0; t < TB; t++) {
                const float *ta = s_tile + (size_t)t * NT * 256;
#pragma unroll
                for (int st = 0; st < 4; st++) {
                    float bv[NBT];
#pragma unroll
                    for (int nb = 0; nb < NBT; nb++) bv[nb] = ta[(NCH + nb) * 256 + (4 * st + q) * 16 + (cl ^ (((2 * st + (q >> 1)) & 3) << 2))];  // B[k = pair][n = co cl]
#pragma unroll
                    for (int ch = 0; ch < NCH; ch++) {
                        const float av = ta[ch * 256 + (4 * st + q) * 16 + (cl ^ (((2 * st + (q >> 1)) & 3) << 2))];  // A[m = ci cl][k = pair 4st + q]
#pragma unroll
                        for (int nb = 0; nb < NBT; nb++)
                            acc[ch * NBT + nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[nb], acc[ch * NBT + nb], 0, 0, 0);
                    }
                }
            }
        }
        if (g_lo == c_lo && g_hi == c_hi) {
            store_slab(k, acc);
        } else if (n_partial == 0 && g_hi < s1) {
            // cut at the FRONT of the slice and more offsets follow: parked in registers until the tiles are free
#pragma unroll
            for (int b = 0; b < NA; b++) part0[b] = acc[b];
            pk0 = k;
            n_partial = 1;
        } else {
            // the slice ends inside this offset: nothing follows, the tile area is free
            float *dst = s_tile + (size_t)n_partial * NA * 256;
#pragma unroll
            for (int b = 0; b < NA; b++)
#pragma unroll
                for (int r = 0; r < 4; r++) dst[b * 256 + r * 64 + l] = acc[b][r];
            if (l == 0) s_pk[w][n_partial] = k;
            n_partial++;  // at most 2: the offsets cut by the two ends of the slice
        }
    }
    if (pk0 >= 0) {
#pragma unroll
        for (int b = 0; b < NA; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) s_tile[b * 256 + r * 64 + l] = part0[b][r];
        if (l == 0) s_pk[w][0] = pk0;
    }
    __syncthreads();
    // offsets cut by slice boundaries (partials summed in wave order) and offsets without pairs in this chunk (zeros)
    for (int k = w; k < p.K; k += nw) {
        const bool empty = s_cum[k + 1] == s_cum[k];
        f32x4 acc[NA];
#pragma unroll
        for (int b = 0; b < NA; b++) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        bool any = false;
        for (int ww = 0; ww < nw; ww++)
            for (int jj = 0; jj < 2; jj++)
                if (s_pk[ww][jj] == k) {
                    any = true;
                    const float *src = lds + (size_t)ww * WAVE_KB * 256 + (size_t)jj * NA * 256;
#pragma unroll
                    for (int b = 0; b < NA; b++)
#pragma unroll
                        for (int r = 0; r < 4; r++) acc[b][r] += src[b * 256 + r * 64 + l];
                }
        if (any || empty) store_slab(k, acc);
    }
}

template <int NBT, int NCH, int TB = WGRAD_TB>
int launch_wgrad_offsetlist(const WgradArgs &p, int nblk_rows, hipStream_t stream)
{
    // per wave: TB x (NCH + NBT) transposition tiles of 1 KB (the boundary partials, 2 x NCH x NBT KB, reuse them)
    constexpr int wave_kb = TB * (NCH + NBT) > 2 * NCH * NBT ? TB * (NCH + NBT) : 2 * NCH * NBT;
    const size_t per_wave = (size_t)wave_kb * 256 * sizeof(float);
    static const int max_w = env_int("MS3D_WGRAD_LIST_WAVES", 16);
    int nw = (int)(LDS_BUDGET / per_wave);
    if (nw > 16) nw = 16;
    if (nw > max_w && max_w >= 1) nw = max_w;
    dim3 grid(nblk_rows, ms3d_divup(p.NBtot, NBT), ms3d_divup(ms3d_divup(p.Cin, 16), NCH));
    const size_t lds = (size_t)nw * per_wave;
    static const hipError_t attr = raise_lds_ceiling((const void *)spconv_wgrad_offsetlist_kernel<NBT, NCH, TB>);
    MS3D_CHECK(attr);
    spconv_wgrad_offsetlist_kernel<NBT, NCH, TB><<<grid, nw * 64, lds, stream>>>(p);
    MS3D_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------ BN helper kernels
// column sums over a [nparts][2][C] partial buffer -> mean / invstd / scale / shift (+ running stats).
// One block per 16 channels; 16 part-lanes per channel sum in double, combined in a fixed order.
__global__ __launch_bounds__(1024) void bn_finalize_stats_kernel(const float *__restrict__ partial, int nparts, int C, long V,
                                                                float eps, float momentum, const float *__restrict__ gamma,
                                                                const float *__restrict__ beta, float *running_mean,
                                                                float *running_var, float *__restrict__ mean_out,
                                                                float *__restrict__ invstd_out, float *__restrict__ scale_out,
                                                                float *__restrict__ shift_out)
{
    __shared__ double s_1[16][17], s_2[16][17];
    const int cl = threadIdx.x & 15, lp = threadIdx.x >> 4;  // 16 columns x 64 part lanes
    const int c = blockIdx.x * 16 + cl;
    double a1 = 0.0, a2 = 0.0;
    if (c < C) {
        // 4 rows per trip, loads issued together (a one-row loop pays one memory round trip per row: 13 us for 1024 rows)
        int p = lp;
        for (; p + 192 < nparts; p += 256) {
            float v1[4], v2[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                v1[u] = partial[((size_t)(p + 64 * u) * 2 + 0) * C + c];
                v2[u] = partial[((size_t)(p + 64 * u) * 2 + 1) * C + c];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                a1 += (double)v1[u];
                a2 += (double)v2[u];
            }
        }
        for (; p < nparts; p += 64) {
            a1 += (double)partial[((size_t)p * 2 + 0) * C + c];
            a2 += (double)partial[((size_t)p * 2 + 1) * C + c];
        }
    }
    // the 4 part lanes of a wave are folded with shuffles (fixed order), then 16 wave rows go through LDS
    a1 += __shfl_xor(a1, 16, 64); a2 += __shfl_xor(a2, 16, 64);
    a1 += __shfl_xor(a1, 32, 64); a2 += __shfl_xor(a2, 32, 64);
    if ((threadIdx.x & 63) < 16) {
        s_1[threadIdx.x >> 6][cl] = a1;
        s_2[threadIdx.x >> 6][cl] = a2;
    }
    __syncthreads();
    if (threadIdx.x >= 16 || c >= C) return;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int i = 0; i < 16; i++) {  // fixed order -> deterministic
        s1 += s_1[i][cl];
        s2 += s_2[i][cl];
    }
    const double mean = s1 / (double)V;
    double var = s2 / (double)V - mean * mean;  // biased, as torch BatchNorm uses for normalisation
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    mean_out[c] = (float)mean;
    invstd_out[c] = invstd;
    scale_out[c] = g * invstd;
    shift_out[c] = b - (float)mean * g * invstd;
    if (running_mean) {
        const double unbiased = (V > 1) ? var * (double)V / (double)(V - 1) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

// per-block partial (sum, sum of squares) per channel.  Round 5, bit-reproducible: a thread owns ONE column group (VEC
// = 4 channels, 16 bytes per lane, when C % 4 == 0; one channel otherwise) of every rpp-th row of the block's row range --
// thread t = (row in pass) * CQ + (column group), so the threads of a pass read one contiguous stretch -- sums in
// registers, parks its sums in LDS, and 2C threads add the rpp copies of their column up in row order.  (Rounds 1-4 added
// into one [2][C] LDS area with float atomics: the order of the additions changed from run to run.)
template <int VEC>
__global__ __launch_bounds__(256) void bn_partial_stats_kernel(const float *__restrict__ x, long V, int C,
                                                               float *__restrict__ partial, int rows_per_block)
{
    __shared__ float s_red[256 * 8];  // [thread][2][VEC]
    const int CQ = C / VEC, rpp = 256 / CQ, tid = threadIdx.x;   // CQ <= 256 (launcher)
    const bool active = tid < rpp * CQ;
    const long r_begin = (long)blockIdx.x * rows_per_block;
    const long r_end = min(V, r_begin + rows_per_block);
    float s1[VEC], s2[VEC];
#pragma unroll
    for (int u = 0; u < VEC; u++) s1[u] = s2[u] = 0.f;
    if (active) {
        const long stride = (long)rpp * CQ;
        const long e_end = r_end * CQ;
        if (VEC == 4) {
            const float4 *x4 = reinterpret_cast<const float4 *>(x);
#pragma unroll 4
            for (long e = r_begin * CQ + tid; e < e_end; e += stride) {
                const float4 t = x4[e];
                s1[0] += t.x; s1[1 % VEC] += t.y; s1[2 % VEC] += t.z; s1[3 % VEC] += t.w;
                s2[0] = fmaf(t.x, t.x, s2[0]); s2[1 % VEC] = fmaf(t.y, t.y, s2[1 % VEC]);
                s2[2 % VEC] = fmaf(t.z, t.z, s2[2 % VEC]); s2[3 % VEC] = fmaf(t.w, t.w, s2[3 % VEC]);
            }
        } else {
#pragma unroll 4
            for (long e = r_begin * CQ + tid; e < e_end; e += stride) {
                const float t = x[e];
                s1[0] += t;
                s2[0] = fmaf(t, t, s2[0]);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < VEC; u++) {
        s_red[tid * 2 * VEC + u] = s1[u];
        s_red[tid * 2 * VEC + VEC + u] = s2[u];
    }
    __syncthreads();
    for (int t = tid; t < 2 * C; t += 256) {
        const int which = t >= C, c = t - which * C, cq = c / VEC, u = c - cq * VEC;
        float sum = 0.f;
        for (int j = 0; j < rpp; j++) sum += s_red[(j * CQ + cq) * 2 * VEC + which * VEC + u];
        partial[(size_t)blockIdx.x * 2 * C + t] = sum;
    }
}

static int launch_bn_partial_stats(const float *x, long V, int C, float *partial, int nblk, int rows_per_block, hipStream_t stream)
{
    if ((C & 3) == 0 && C <= 1024)
        bn_partial_stats_kernel<4><<<nblk, 256, 0, stream>>>(x, V, C, partial, rows_per_block);
    else if (C <= 256)
        bn_partial_stats_kernel<1><<<nblk, 256, 0, stream>>>(x, V, C, partial, rows_per_block);
    else
        return MS3D_E_UNSUPPORTED;
    MS3D_LAUNCH_CHECK();
    return 0;
}

// y = x*scale + shift (optionally ReLU): the stand-alone BN(+ReLU) at the end of the U-Net
__global__ void bn_apply_kernel(const float *__restrict__ x, long n, int C, const float *__restrict__ scale,
                                const float *__restrict__ shift, int relu, float *__restrict__ y)
{
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        float v = fmaf(x[e], scale[c], shift ? shift[c] : 0.f);
        y[e] = relu ? fmaxf(v, 0.f) : v;
    }
}

// dx = scale * (dz - s1/V - xhat * s2/V)   (in place on dz allowed); s1 = sum dz, s2 = sum dz*xhat
__global__ void bn_bwd_apply_kernel(const float *__restrict__ dz, const float *__restrict__ x, long n, int C, long V,
                                    const float *__restrict__ scale, const float *__restrict__ mean,
                                    const float *__restrict__ invstd, const float *__restrict__ s1s2,
                                    const float *__restrict__ add /* or null: a gradient arriving over a skip connection */,
                                    float *__restrict__ dx)
{
    const float invV = 1.f / (float)V;
    if ((C & 3) == 0) {
        // 16 bytes per lane, one channel-index computation per four elements
        const long n4 = n >> 2;
        const int C4 = C >> 2;
        for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < n4; v += (long)gridDim.x * blockDim.x) {
            const int c = (int)(v % C4) * 4;
            const float4 xv = reinterpret_cast<const float4 *>(x)[v], g = reinterpret_cast<const float4 *>(dz)[v];
            const float4 mu = *reinterpret_cast<const float4 *>(mean + c), is = *reinterpret_cast<const float4 *>(invstd + c);
            const float4 sc = *reinterpret_cast<const float4 *>(scale + c);
            const float4 a1 = *reinterpret_cast<const float4 *>(s1s2 + c), a2 = *reinterpret_cast<const float4 *>(s1s2 + C + c);
            float4 o;
            o.x = sc.x * (g.x - a1.x * invV - ((xv.x - mu.x) * is.x) * a2.x * invV);
            o.y = sc.y * (g.y - a1.y * invV - ((xv.y - mu.y) * is.y) * a2.y * invV);
            o.z = sc.z * (g.z - a1.z * invV - ((xv.z - mu.z) * is.z) * a2.z * invV);
            o.w = sc.w * (g.w - a1.w * invV - ((xv.w - mu.w) * is.w) * a2.w * invV);
            if (add) {
                const float4 r = reinterpret_cast<const float4 *>(add)[v];
                o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
            }
            reinterpret_cast<float4 *>(dx)[v] = o;
        }
        return;
    }
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const float xh = (x[e] - mean[c]) * invstd[c];
        const float o = scale[c] * (dz[e] - s1s2[c] * invV - xh * s1s2[C + c] * invV);
        dx[e] = add ? o + add[e] : o;
    }
}

// stand-alone BN(+ReLU) backward, stage 1: dz = dy * [relu mask]; partial sums of dz and dz*xhat.  Thread layout and the
// fixed-order combination of bn_partial_stats_kernel (bit-reproducible; sums in registers -- two LDS atomics per ELEMENT
// made the first build of this kernel 133 us for 433k x 32 rows, 7x its traffic).
template <int VEC>
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float *__restrict__ dy, const float *__restrict__ x,
                                                             long V, int C, const float *__restrict__ scale,
                                                             const float *__restrict__ shift,
                                                             const float *__restrict__ mean,
                                                             const float *__restrict__ invstd, int relu,
                                                             float *__restrict__ dz, float *__restrict__ partial,
                                                             int rows_per_block)
{
    __shared__ float s_red[256 * 8];  // [thread][2][VEC]
    const int CQ = C / VEC, rpp = 256 / CQ, tid = threadIdx.x;
    const bool active = tid < rpp * CQ;
    const long r_begin = (long)blockIdx.x * rows_per_block;
    const long r_end = min(V, r_begin + rows_per_block);
    float s1[VEC], s2[VEC];
#pragma unroll
    for (int u = 0; u < VEC; u++) s1[u] = s2[u] = 0.f;
    if (active) {
        const int c = (tid % CQ) * VEC;
        const long stride = (long)rpp * CQ, e_end = r_end * CQ;
        if (VEC == 4) {
            const float4 sc = *reinterpret_cast<const float4 *>(scale + c), sh = *reinterpret_cast<const float4 *>(shift + c);
            const float4 mu = *reinterpret_cast<const float4 *>(mean + c), is = *reinterpret_cast<const float4 *>(invstd + c);
#pragma unroll 2
            for (long e = r_begin * CQ + tid; e < e_end; e += stride) {
                const float4 xv = reinterpret_cast<const float4 *>(x)[e];
                float4 g = reinterpret_cast<const float4 *>(dy)[e];
                if (relu) {
                    if (!(fmaf(xv.x, sc.x, sh.x) > 0.f)) g.x = 0.f;
                    if (!(fmaf(xv.y, sc.y, sh.y) > 0.f)) g.y = 0.f;
                    if (!(fmaf(xv.z, sc.z, sh.z) > 0.f)) g.z = 0.f;
                    if (!(fmaf(xv.w, sc.w, sh.w) > 0.f)) g.w = 0.f;
                }
                reinterpret_cast<float4 *>(dz)[e] = g;
                s1[0] += g.x; s1[1 % VEC] += g.y; s1[2 % VEC] += g.z; s1[3 % VEC] += g.w;
                s2[0] += g.x * ((xv.x - mu.x) * is.x); s2[1 % VEC] += g.y * ((xv.y - mu.y) * is.y);
                s2[2 % VEC] += g.z * ((xv.z - mu.z) * is.z); s2[3 % VEC] += g.w * ((xv.w - mu.w) * is.w);
            }
        } else {
            const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
#pragma unroll 2
            for (long e = r_begin * CQ + tid; e < e_end; e += stride) {
                const float xv = x[e];
                float g = dy[e];
                if (relu && !(fmaf(xv, sc, sh) > 0.f)) g = 0.f;
                dz[e] = g;
                s1[0] += g;
                s2[0] += g * ((xv - mu) * is);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < VEC; u++) {
        s_red[tid * 2 * VEC + u] = s1[u];
        s_red[tid * 2 * VEC + VEC + u] = s2[u];
    }
    __syncthreads();
    for (int t = tid; t < 2 * C; t += 256) {
        const int which = t >= C, c = t - which * C, cq = c / VEC, u = c - cq * VEC;
        float sum = 0.f;
        for (int j = 0; j < rpp; j++) sum += s_red[(j * CQ + cq) * 2 * VEC + which * VEC + u];
        partial[(size_t)blockIdx.x * 2 * C + t] = sum;
    }
}

// out[t] = sum_p partial[p][t]: 16 columns x 16 part-lanes per block, fixed combination order
__global__ __launch_bounds__(1024) void reduce_partial_kernel(const float *__restrict__ partial, int nparts, int n,
                                                              float *__restrict__ out)
{
    __shared__ double s_sum[16][17];
    const int col = blockIdx.x * 16 + (threadIdx.x & 15), lane_p = threadIdx.x >> 4;
    double s = 0.0;
    if (col < n)
        for (int p = lane_p; p < nparts; p += 64) s += (double)partial[(size_t)p * n + col];
    // fold the 4 part lanes of a wave with shuffles, then 16 wave rows through LDS (fixed order)
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if ((threadIdx.x & 63) < 16) s_sum[threadIdx.x >> 6][threadIdx.x & 15] = s;
    __syncthreads();
    if (threadIdx.x < 16 && col < n) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 16; i++) t += s_sum[i][threadIdx.x];
        out[col] = (float)t;
    }
}

// reduce_partial_kernel + bn_bwd_apply_kernel in ONE launch (the BatchNorm-backward chain behind every fused backward-data
// convolution: ~80 pairs per training step).  Workgroups take tickets in the order they start; the first ceil(2C / 16) of
// them reduce 16 columns of the partials each -- the arithmetic of reduce_partial_kernel, same lanes, same order -- and
// PUBLISH the sums as 64-bit words (flag | float bits) with relaxed agent-scope atomic stores (the payload rides in the
// word: no fence, no L2 write-back; the XCDs' L2s are not coherent with each other); every workgroup then collects the
// 2C words (spinning on a word only until its reducer, which holds a lower ticket and is therefore running or done, has
// stored it) and applies dx = scale * (dz - s1/V - xhat * s2/V) [+ add] to its share of the rows with the expression of
// bn_bwd_apply_kernel.  The workgroup that finishes last clears the words for the next launch on the stream.
struct BnBwdState {
    unsigned long long word[1024];   // 2C <= 1024
    int ticket, done;
};

__global__ __launch_bounds__(1024) void bn_bwd_reduce_apply_kernel(const float *__restrict__ partial, int nparts, const float *dz,
                                                                  const float *__restrict__ x, long n, int C, long V,
                                                                  const float *__restrict__ scale,
                                                                  const float *__restrict__ mean,
                                                                  const float *__restrict__ invstd,
                                                                  const float *__restrict__ add, float *dx,
                                                                  float *__restrict__ s1s2_out, BnBwdState *__restrict__ st)
{
    __shared__ double s_sum[16][17];
    __shared__ float s_s[1024];
    __shared__ int s_bid;
    const int n2 = 2 * C, nred = (n2 + 15) / 16;
    if (threadIdx.x == 0) s_bid = atomicAdd(&st->ticket, 1);
    __syncthreads();
    const int bid = s_bid;
    if (bid < nred) {
        const int col = bid * 16 + (threadIdx.x & 15), lane_p = threadIdx.x >> 4;
        double s = 0.0;
        if (col < n2)
            for (int p = lane_p; p < nparts; p += 64) s += (double)partial[(size_t)p * n2 + col];
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if ((threadIdx.x & 63) < 16) s_sum[threadIdx.x >> 6][threadIdx.x & 15] = s;
        __syncthreads();
        if (threadIdx.x < 16 && col < n2) {
            double t = 0.0;
#pragma unroll
            for (int i = 0; i < 16; i++) t += s_sum[i][threadIdx.x];
            const float f = (float)t;
            s1s2_out[col] = f;
            __hip_atomic_store(&st->word[col], (1ull << 32) | (unsigned long long)__float_as_uint(f), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    for (int t = threadIdx.x; t < n2; t += 1024) {
        unsigned long long w;
        do {
            w = __hip_atomic_load(&st->word[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } while ((w >> 32) == 0);
        s_s[t] = __uint_as_float((unsigned)w);
    }
    __syncthreads();
    if (dx) {
        const float invV = 1.f / (float)V;
        const long stride = (long)gridDim.x * 1024;
        if ((C & 3) == 0) {
            const long n4 = n >> 2;
            const int C4 = C >> 2;
            for (long v = (long)bid * 1024 + threadIdx.x; v < n4; v += stride) {
                const int c = (int)(v % C4) * 4;
                const float4 xv = reinterpret_cast<const float4 *>(x)[v], g = reinterpret_cast<const float4 *>(dz)[v];
                const float4 mu = *reinterpret_cast<const float4 *>(mean + c), is = *reinterpret_cast<const float4 *>(invstd + c);
                const float4 sc = *reinterpret_cast<const float4 *>(scale + c);
                const float4 a1 = *reinterpret_cast<const float4 *>(s_s + c), a2 = *reinterpret_cast<const float4 *>(s_s + C + c);
                float4 o;
                o.x = sc.x * (g.x - a1.x * invV - ((xv.x - mu.x) * is.x) * a2.x * invV);
                o.y = sc.y * (g.y - a1.y * invV - ((xv.y - mu.y) * is.y) * a2.y * invV);
                o.z = sc.z * (g.z - a1.z * invV - ((xv.z - mu.z) * is.z) * a2.z * invV);
                o.w = sc.w * (g.w - a1.w * invV - ((xv.w - mu.w) * is.w) * a2.w * invV);
                if (add) {
                    const float4 r = reinterpret_cast<const float4 *>(add)[v];
                    o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
                }
                reinterpret_cast<float4 *>(dx)[v] = o;
            }
        } else {
            for (long e = (long)bid * 1024 + threadIdx.x; e < n; e += stride) {
                const int c = (int)(e % C);
                const float xh = (x[e] - mean[c]) * invstd[c];
                const float o = scale[c] * (dz[e] - s_s[c] * invV - xh * s_s[C + c] * invV);
                dx[e] = add ? o + add[e] : o;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(&st->done, 1) == (int)gridDim.x - 1) {
        for (int t = 0; t < n2; t++) __hip_atomic_store(&st->word[t], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&st->ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&st->done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

BnBwdState *bn_bwd_state(hipStream_t stream)
{
    // keyed by (device, stream): the null stream's handle is the same on every device
    static std::mutex lock;
    static std::map<std::pair<int, hipStream_t>, BnBwdState *> states;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> guard(lock);
    auto it = states.find({dev, stream});
    if (it != states.end()) return it->second;
    BnBwdState *p = nullptr;
    if (hipMalloc((void **)&p, sizeof(BnBwdState)) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, sizeof(BnBwdState)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return nullptr;
    states.emplace(std::make_pair(dev, stream), p);
    return p;
}

}  // namespace

extern "C" {

size_t ms3d_spconv_wf_floats(int K, int Cin, int Cout)
{
    return (size_t)K * ms3d_divup(Cin, 16) * 4 * ms3d_divup(Cout, 16) * 64;
}

int ms3d_spconv_prep_weights(const float *W, int K, int Cin_eff, int Cout_eff, int transpose, int mirror, float *wf,
                             float *wf_stream, ms3d_stream_t stream)
{
    const int NCH = ms3d_divup(Cin_eff, 16), NBtot = ms3d_divup(Cout_eff, 16);
    const long total = (long)K * NCH * 4 * NBtot * 64;
    prep_weights_kernel<<<(int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048), 256, 0, (hipStream_t)stream>>>(
        W, wf, K, Cin_eff, Cout_eff, NCH, NBtot, transpose, mirror, nullptr, 0, 0, 0, wf_stream, nullptr, 1);
    MS3D_LAUNCH_CHECK();
    return 0;
}

// forward image wf (W[K][Cin][Cout]) and backward-data image wft (W^T, optionally offset-mirrored) in ONE launch
static int prep_weights_pair_impl(const float *W, int K, int Cin, int Cout, int mirror_bwd, float *wf, float *wft,
                                  float *wf_aux, float *wft_aux, int aux_kind, hipStream_t stream)
{
    const int NCH = ms3d_divup(Cin, 16), NBtot = ms3d_divup(Cout, 16);
    const int NCH2 = ms3d_divup(Cout, 16), NBtot2 = ms3d_divup(Cin, 16);
    const long total = (long)K * NCH * 4 * NBtot * 64, total2 = (long)K * NCH2 * 4 * NBtot2 * 64;
    const long m = total > total2 ? total : total2;
    prep_weights_kernel<<<(int)((m + 255) / 256 < 2048 ? (m + 255) / 256 : 2048), 256, 0, stream>>>(
        W, wf, K, Cin, Cout, NCH, NBtot, 0, 0, wft, NCH2, NBtot2, mirror_bwd, wf_aux, wft_aux, aux_kind);
    MS3D_LAUNCH_CHECK();
    return 0;
}
int ms3d_spconv_prep_weights_pair(const float *W, int K, int Cin, int Cout, int mirror_bwd, float *wf, float *wft,
                                  float *wf_stream, float *wft_stream, ms3d_stream_t stream)
{
    return prep_weights_pair_impl(W, K, Cin, Cout, mirror_bwd, wf, wft, wf_stream, wft_stream, 1, (hipStream_t)stream);
}

// the same for n layers in one launch; descs = device array of {W, wf, wft, K, Cin, Cout, mirror_bwd, block_begin, 0}
// (48 bytes each, block_begin = running sum of ms3d_spconv_prep_blocks), total_blocks = sum over the layers
int ms3d_spconv_prep_blocks(int K, int Cin, int Cout)
{
    return (int)(((long)K * ms3d_divup(Cin, 16) * 4 * ms3d_divup(Cout, 16) * 64 + 255) / 256);
}
int ms3d_spconv_prep_weights_multi(const void *descs, int n, int total_blocks, ms3d_stream_t stream)
{
    if (n <= 0 || total_blocks <= 0) return 0;
    prep_weights_multi_kernel<<<total_blocks, 256, 0, (hipStream_t)stream>>>(reinterpret_cast<const PrepDesc *>(descs), n);
    MS3D_LAUNCH_CHECK();
    return 0;
}

// mean / invstd / scale / shift (+ running stats) from per-block (sum, sum of squares) partials written by a conv epilogue
int ms3d_bn_finalize(const float *partial, int nparts, long V, int C, float eps, float momentum, const float *gamma,
                     const float *beta, float *running_mean, float *running_var, float *mean, float *invstd,
                     float *scale, float *shift, ms3d_stream_t stream);

}  // extern "C"

namespace {
bool bf3_enabled();
struct FwdGeom {
    int rt;   // small levels on the bf16x3 image: 16-row tiles per block (1 = the one-tile kernels)
    int nbt, ny, threads, nblk, G;
    size_t lds;
    bool ok, small, pairlist, stream;
    int cg;   // stream kernel: 16-channel chunks per weight group
    int pl_rows;  // pair-list kernel: rows per tile of the list it walks (64, or 32 for 32 -> 32 layers)
    bool ws;  // weight-stationary kernel of the coarse levels: ws_ng offset groups of ws_kg, ws_R row parts of ws_tpp tiles
    int ws_ng, ws_kg, ws_R, ws_tpp;
};
// Weight-stationary route (spconv_fwd_ws_kernel): levels of at most MS3D_WS_MAX_TILES 16-row tiles (default 220 = 3.5k rows;
// 0 switches the route off), 48+ channels on both sides, K > 1.  The geometry must not depend on anything but the layer
// shape: ms3d_spconv_partial_blocks sizes the statistics partials from it before the launch.
bool ws_geometry(int Vout, int K, int Cin, int Cout, FwdGeom &g)
{
    static const int max_tiles = env_int("MS3D_WS_MAX_TILES", 220);
    static const int env_nbt = env_int("MS3D_WS_NBT", 0);
    static const int lds_kb = env_int("MS3D_WS_LDS_KB", 128);
    static const int target_blocks = env_int("MS3D_WS_BLOCKS", 1024);
    static const int min_tpp = env_int("MS3D_WS_MIN_TPP", 4);
    // MS3D_WS_ALL=1: every shape the kernel can serve (tests, measurements).  Default: only where it was measured to win --
    // K = 27 layers with a side beyond 256 channels (no three-piece bf16 image exists for them: 320 -> 160 and its
    // backward-data twin at the 2.5k-row level of the m = 32 models) on levels of 64+ tiles.  Measured, us per launch, this
    // route | one-tile kernels (profiles/r06_ws_*.txt): 320 -> 160 at 2.5k rows 136 | 162; but 160 -> 160 79 | 87 (f32) | 60
    // (bf16x3), 80 -> 80 32 | 26, 192 -> 192 at 493 rows 35 | 31, 384 -> 192 78 | 58, 96 -> 96 17.5 | 12.5, 224 -> 224 at
    // 112 rows 22.6 | 21.0: the one-tile kernels are NOT bound by re-streaming the weights (L2 hit rate 94 %, 134-cycle
    // average L2 latency, profiles/r06_conv_counters.txt) and this route pays a staging round trip, a release fence and a
    // combine pass on top of the same f32 MFMA time.
    static const int ws_all = env_int("MS3D_WS_ALL", 0);
    const int NCH = ms3d_divup(Cin, 16), NBtot = ms3d_divup(Cout, 16), ntiles = ms3d_divup(Vout, 16);
    if (max_tiles <= 0 || ntiles > max_tiles || K < 2 || K > 27 || Cin % 16 || Cout % 16 || Cin < 48 || Cout < 48 || Cin > 512) return false;
    if (!ws_all && !(K == 27 && (Cin > 256 || Cout > 256) && Cin >= 160 && ntiles >= 64)) return false;
    int nbt = NBtot % 2 == 0 ? 2 : (NBtot % 3 == 0 ? 3 : 1);
    if (env_nbt >= 1 && env_nbt <= 4 && NBtot % env_nbt == 0) nbt = env_nbt;
    static const int env_waves = env_int("MS3D_WS_WAVES", 16);
    const int waves = env_waves < 1 ? 1 : (env_waves > 16 ? 16 : env_waves);
    for (;; ) {
        const size_t per_offset = (size_t)NCH * 4 * nbt * 64 * sizeof(float);
        const size_t fixed = (size_t)2 * NCH * 16 * sizeof(float) + (size_t)waves * 12 * 16 * sizeof(int) +
                             waves * stat_slot_floats(nbt) * sizeof(float) + 16;
        int kg = (int)(((size_t)lds_kb * 1024 - fixed) / per_offset);
        if (kg < 1) {
            if (nbt == 1) return false;
            nbt = 1;                       // very wide input side: one column block per slice
            continue;
        }
        if (kg > 12) kg = 12;             // table entries per tile: three per lane
        if (kg > K) kg = K;
        const int ng = ms3d_divup(K, kg);
        kg = ms3d_divup(K, ng);
        const int ny = NBtot / nbt;
        int R = target_blocks / (ng * ny);
        if (R < 1) R = 1;
        int tpp = ms3d_divup(ntiles, R);
        if (tpp < min_tpp) tpp = min_tpp;
        R = ms3d_divup(ntiles, tpp);
        g = FwdGeom{};
        g.ws = true;
        g.ok = true;
        g.nbt = nbt; g.ny = ny; g.threads = waves * 64;
        g.nblk = R;                        // statistics partial rows = R x ny (one per unit)
        g.G = K; g.rt = 1;
        g.ws_ng = ng; g.ws_kg = kg; g.ws_R = R; g.ws_tpp = tpp;
        g.lds = per_offset * kg + (size_t)2 * NCH * 16 * sizeof(float) + (size_t)waves * kg * 16 * sizeof(int) +
                waves * stat_slot_floats(nbt) * sizeof(float) + 16;
        return true;
    }
}
// slab area + arrival counters of the weight-stationary kernels, one per (device, stream), grown on demand (a launch on a
// stream is ordered behind the previous one, so one area per stream is enough); counters are zero between launches
struct WsSpace { float *slabs = nullptr; size_t slab_floats = 0; unsigned *cnt = nullptr; int ncnt = 0; };
WsSpace *ws_space(hipStream_t stream, size_t need_floats, int need_cnt)
{
    static std::mutex lock;
    static std::map<std::pair<int, hipStream_t>, WsSpace> spaces;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> guard(lock);
    WsSpace &sp = spaces[{dev, stream}];
    if (need_floats > sp.slab_floats) {
        if (sp.slabs) { if (hipStreamSynchronize(stream) != hipSuccess || hipFree(sp.slabs) != hipSuccess) return nullptr; }
        sp.slabs = nullptr; sp.slab_floats = 0;
        const size_t n = need_floats < ((size_t)8 << 20) ? ((size_t)8 << 20) : need_floats + need_floats / 2;   // >= 32 MB
        if (hipMalloc((void **)&sp.slabs, n * sizeof(float)) != hipSuccess) return nullptr;
        sp.slab_floats = n;
    }
    if (need_cnt > sp.ncnt) {
        if (sp.cnt) { if (hipStreamSynchronize(stream) != hipSuccess || hipFree(sp.cnt) != hipSuccess) return nullptr; }
        sp.cnt = nullptr; sp.ncnt = 0;
        const int n = need_cnt < 4096 ? 4096 : 2 * need_cnt;
        if (hipMalloc((void **)&sp.cnt, (size_t)n * sizeof(unsigned)) != hipSuccess) return nullptr;
        if (hipMemset(sp.cnt, 0, (size_t)n * sizeof(unsigned)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return nullptr;
        sp.ncnt = n;
    }
    return &sp;
}
constexpr int PAIRLIST_MIN_ROWS = 30000;  // below this the 16-row kernels win (weight staging per block dominates)
int pairlist_min_rows()
{
    static const int v = [] {
        const char *e = getenv("MS3D_PAIRLIST_MIN_ROWS");  // tuning knob; < 0 disables the pair-list kernels
        return e ? atoi(e) : PAIRLIST_MIN_ROWS;
    }();
    return v;
}
int pairstream_mode()
{
    // 1 = rectangular layers above 32 channels stream their weights from L2 (spconv_fwd_pairstream_kernel); 3 = all
    // layers above 32 channels do; 2 = up to 64 input channels take the LDS-resident pair-list kernel in 16-column
    // slices instead; 0 = table walk
    static const int v = [] { const char *e = getenv("MS3D_PAIRSTREAM"); return e ? atoi(e) : 1; }();
    return v;
}
bool pairlist_shape_ok(int Vout, int K, int Cin, int Cout)
{
    // Vout <= 2^22: the kernel addresses input rows with 32-bit element offsets (Vin <= 8 * Vout for a stride-2 map)
    const bool wide_ok = pairstream_mode() == 2 && Cin <= 64 && Cout <= 256 && (size_t)Vout * 8 * Cin < (1ull << 32);
    return pairlist_min_rows() >= 0 && Vout >= pairlist_min_rows() && Vout <= (1 << 22) && K > 1 && K <= 27 &&
           Cin % 16 == 0 && Cout % 16 == 0 && ((Cin <= 32 && Cout <= 32) || wide_ok);
}
// wide layers (a side above 32 channels) on pair-listed tables: weights streamed from L2 (spconv_fwd_pairstream_kernel).
// Measured on the benchmark's tables (tools/conv_micro.py, us per launch, stream vs table walk): 64 -> 32 at level 0
// 221 vs 417, 32 -> 64 238 vs 388, 128 -> 64 at level 1 647 vs 795, 64 -> 128 520 vs 940 -- but 64 -> 64 at level 1
// 360 vs 384, 48 -> 48 at level 2 108 vs 85, 96 -> 96 326 vs 295: the square layers of the deeper levels have 10-12
// of 27 neighbours per row and the table walk skips a tile's empty offsets, the kernel here pays ~340 non-MFMA
// instructions and five dependent LDS / memory waits per 64-MFMA stage (SQ counters: 60 % of the wave cycles in issue
// stalls) and does not overlap them with the matrix pipe.  So it takes the RECTANGULAR layers (the first convolution
// after a concatenation and its backward-data twin), where the table walk's column-slice geometry is at its worst.
// the part of the decision that depends on the layer only (the weight layout kernels write the streamed images for
// exactly these layers)
bool bf3_enabled()
{
    static const bool on = [] { const char *e = getenv("MS3D_BF16X3"); return !e || atoi(e) != 0; }();
    return on;
}
bool pairstream_layer_ok(int K, int Cin, int Cout)
{
    // layers with both sides >= 64 channels (multiples of 32) run the table walk on three-piece bf16 operands instead:
    // 128 -> 64 at level 1 506 vs 755 us, 64 -> 128 415 vs 779, 192 -> 96 at level 2 391 vs 641, 64 -> 96 (K = 8) 43 vs 81
    if (bf3_enabled() && bf3_dims_ok(K, Cin, Cout)) return false;
    const int mode = pairstream_mode();   // 3 = every wide layer (experiments)
    const bool on = (mode == 1 && Cin != Cout) || mode == 3 || (mode == 2 && Cin > 64);
    return on && pairlist_min_rows() >= 0 && K > 1 && K <= 27 && Cin % 16 == 0 && Cout % 16 == 0 && (Cin > 32 || Cout > 32) &&
           Cin <= 256 && Cout <= 256;
}
bool pairstream_shape_ok(int Vout, int K, int Cin, int Cout)
{
    return pairstream_layer_ok(K, Cin, Cout) && Vout >= pairlist_min_rows() && Vout <= (1 << 22);
}
constexpr int SMALL_TILES = 1100;  // <= ~17k output rows: direct-B split-K kernel (measured faster than LDS staging up to here)
// Launch geometry shared by the launcher and ms3d_spconv_partial_blocks.  Small levels (a few hundred rows at the
// bottom of the U-Net) are spread over the chip by giving each wave fewer output columns and each block fewer waves.
// pl_rows: rows per tile of the pair list the call is given (0 = none)
FwdGeom fwd_geometry(int Vout, int K, int Cin, int Cout, bool with_bn_partial, int pl_rows)
{
    const bool with_pairlist = pl_rows != 0;
    FwdGeom g{};
    const int NCH = ms3d_divup(Cin, 16), NBtot = ms3d_divup(Cout, 16);
    const int ntiles = ms3d_divup(Vout, 16);
    static const int small_tiles = [] {
        const char *e = getenv("MS3D_SMALL_TILES");  // tuning knob; default measured on MI355X
        return e ? atoi(e) : SMALL_TILES;
    }();
    if (ws_geometry(Vout, K, Cin, Cout, g)) return g;
    if (ntiles <= small_tiles && (size_t)NCH * NBtot >= 4) {
        // one block per (tile, column slice); waves = offset groups; enough column splits for ~1024+ waves
        const int ks = ms3d_divup(K, OG);
        int ny = ms3d_divup(NBtot, MAX_NBT);
        while (NBtot % ny != 0) ny++;
        while ((long)ntiles * ny * ks < 1024 && ny < NBtot) {
            ny++;
            while (NBtot % ny != 0) ny++;
        }
        g.small = true;
        g.ny = ny;
        g.nbt = NBtot / ny;
        g.threads = ks * 64;
        g.nblk = ntiles;
        g.G = K;
        g.rt = 1;
        g.lds = ((size_t)(ks - 1) * g.nbt * 256 + (with_bn_partial ? stat_slot_floats(g.nbt) : 0)) * sizeof(float) + 16;
        g.ok = ks <= 4 && g.nbt >= 1 && g.nbt <= MAX_NBT;
        // bf16x3 layers with three or more rounds of one-tile blocks: three tiles per block share every weight load
        // (spconv_fwd_small_bf3_rt_kernel) and one block per CU remains.  Measured at 11.7k rows, us per launch with 1 / 2 /
        // 3 tiles per block: 128 -> 128 140 / 153 / 91, 96 -> 96 78 / 100 / 60, 112 -> 112 121 / 138 / 80, 64 -> 128
        // 74 / 82 / 49, 64 -> 64 40 / 41 / 38 (two tiles leave 1.5 rounds of blocks); levels with fewer tiles (160 -> 160 at
        // 2.5k rows, five column slices) lose: 60 / 73 / 69.
        static const int env_rt = [] { const char *e = getenv("MS3D_SMALL_RT"); return e ? atoi(e) : 3; }();
        if (bf3_enabled() && bf3_dims_ok(K, Cin, Cout) && g.nbt >= 2 && K >= 4) {
            const int rt = (ntiles >= 3 * 230 && env_rt >= 3) ? 3 : 1;
            if (rt > 1) {
                const int waves = 4;
                g.rt = rt;
                g.threads = waves * 64;
                g.nblk = ms3d_divup(ntiles, rt);
                g.lds = ((size_t)waves * rt * g.nbt * 256 + (with_bn_partial ? waves * stat_slot_floats(g.nbt) : 0)) * sizeof(float) + 16;
            }
        }
        return g;
    }
    if (with_pairlist && pairstream_shape_ok(Vout, K, Cin, Cout)) {
        // Column slices of at most two 16-column blocks: a 128-row accumulator tile is then <= 19 KB, EIGHT waves fit a CU
        // -- two per SIMD, so one wave's loads / LDS updates / register shuffling issue under the other's MFMAs (with one
        // wave per SIMD and 64-column tiles the compiler's straight-line MFMA blocks left the rest un-overlapped: 360 us
        // at 64 -> 64 against 150 us of matrix-pipe time) -- at the price of gathering the input rows once per slice.
        static const int env_nbt = [] { const char *e = getenv("MS3D_PS_NBT"); return e ? atoi(e) : 0; }();
        static const int env_w = [] { const char *e = getenv("MS3D_PS_W"); return e ? atoi(e) : 8; }();
        int nbt = 1, cg = 1;
        // Round 6: K = 27 layers whose column-block count is a multiple of four take all four blocks in one wave (the rows are
        // gathered once instead of once per two-block slice; eight 35 KB tiles do not fit, the launch runs 4 waves): 32 -> 64
        // at 196k rows 273 -> 218 us, the backward-data side of 64 -> 32 284 -> 254; three-block slices lose (32 -> 48, K = 8:
        // 34.7 -> 44.2) and keep the two-block rule.  MS3D_PS_NBT=n: at most n blocks per wave, as before.
        const int max_nbt = env_nbt > 0 ? env_nbt : ((K == 27 && NBtot % 4 == 0) ? 4 : 2);
        for (int c = max_nbt; c >= 1; c--)
            if (NBtot % c == 0) { nbt = c; break; }
        for (int c = 4; c >= 1; c--)
            if (NCH % c == 0) { cg = c; break; }
        const int ny = NBtot / nbt;
        const size_t spart = (size_t)(((2 * Cout + 3) & ~3) + 4 + ((2 * Cin + 3) & ~3)) * sizeof(float);
        const size_t perwave = pairstream_wave_floats(nbt) * sizeof(float);
        int W = env_w;
        while (W > 1 && spart + (size_t)W * perwave > 158 * 1024) W--;
        const int tiles = ms3d_divup(Vout, PSR);
        const int wfill = ms3d_divup(tiles, MS3D_PL_PARTS);   // no more waves than a part has tiles
        if (W > wfill) W = wfill < 1 ? 1 : wfill;
        g.stream = true;
        g.cg = cg;
        g.ny = ny;
        g.nbt = nbt;
        g.threads = W * 64;
        g.nblk = MS3D_PL_PARTS;
        g.G = K;
        g.lds = spart + (size_t)W * perwave;
        g.ok = true;
        return g;
    }
    if (with_pairlist && pairlist_shape_ok(Vout, K, Cin, Cout)) {
        // column split (the gathers are repeated per slice) only when the weight image would starve the block of waves
        int nbt = NBtot;
        const size_t spart = (size_t)(((2 * Cout + 3) & ~3) + 4) * sizeof(float);  // statistics + the tile pick counter
        static const int env_w = [] {
            const char *e = getenv("MS3D_PL_W");  // tuning knob: cap on waves per block
            return e ? atoi(e) : 0;
        }();
        int cr = MS3D_PL_ROWS;
        auto waves_for = [&](int nbt_) {
            const size_t wbytes = (size_t)K * NCH * 4 * nbt_ * 64 * sizeof(float);
            const size_t perwave = pairlist_wave_floats(nbt_, cr) * sizeof(float);
            return wbytes + spart >= LDS_BUDGET ? 0 : (int)((LDS_BUDGET - wbytes - spart) / perwave);
        };
        static const int min_waves = [] { const char *e = getenv("MS3D_PL_MIN_WAVES"); return e ? atoi(e) : 8; }();
        // 32 -> 32 (K = 27): both column blocks in one wave on 32-row tiles when 64-row tiles leave fewer than min_waves --
        // taken when the call comes with a 32-row list (pl_rows), which the caller builds for DENSE tables only
        // (ms3d_spconv_pairlist_rows_dense).  Measured (profiles/r06_pairlist_narrow.txt, us per launch, two 16-column
        // slices on 64-row tiles | narrow): level 1 of the bench batch (196k rows, 10.2 of 27 neighbours per row) forward
        // 98.9 | 88.8, backward-data side 126.9 | 111.6, and -0.35 ms of convolution time per PointGroup step; level 0
        // (417k rows, 5.5 neighbours: where the m = 32 models have their 32-channel layers) 120.8 | 133.7 -- a 32-row
        // tile has 6.5 pairs per offset there and pads them to 16.
        if (pl_rows == PL_ROWS_NARROW) {
            cr = PL_ROWS_NARROW;
            if (!(nbt == 2 && NCH == 2 && waves_for(nbt) >= min_waves)) { g.ok = false; return g; }   // not a list this shape can walk
        }
        while (nbt > 1 && waves_for(nbt) < min_waves) {   // next smaller divisor of the column-block count
            int d = nbt - 1;
            while (NBtot % d != 0) d--;
            nbt = d;
        }
        int W = waves_for(nbt);
        if (cr == PL_ROWS_NARROW && W > 10) W = 10;     // the narrow kernel's launch bound
        const int tiles = ms3d_divup(Vout, cr);
        g.pl_rows = cr;
        if (W > 16) W = 16;
        if (env_w > 0 && W > env_w) W = env_w;
        if (W >= 2) {
            // one block per part of the list (MS3D_PL_PARTS = one per CU); no more waves than a part has tiles
            const int wfill = ms3d_divup(tiles, MS3D_PL_PARTS);
            if (W > wfill) W = wfill < 2 ? 2 : wfill;
            g.pairlist = true;
            g.ny = NBtot / nbt;
            g.nbt = nbt;
            g.threads = W * 64;
            g.nblk = MS3D_PL_PARTS;
            g.G = K;
            g.lds = (size_t)K * NCH * 4 * nbt * 64 * sizeof(float) + spart + (size_t)W * pairlist_wave_floats(nbt, cr) * sizeof(float);
            g.ok = true;
            return g;
        }
    }
    int ny = ms3d_divup(NBtot, MAX_NBT);
    while (NBtot % ny != 0) ny++;
    while ((long)ntiles * ny < 1024 && ny < NBtot) {  // more column splits until ~4 waves per CU exist
        ny++;
        while (NBtot % ny != 0) ny++;
    }
    g.ny = ny;
    g.nbt = NBtot / ny;
    const size_t per_offset = (size_t)NCH * 4 * g.nbt * 64 * sizeof(float);  // LDS bytes per offset (column slice)
    const size_t extra = with_bn_partial ? STAT_MAX_WAVES * stat_slot_floats(g.nbt) * sizeof(float) : 0;  // a statistics slot per wave
    const bool resident = per_offset * K + extra <= LDS_BUDGET;
    if (resident) {
        g.G = K;
        const size_t need = per_offset * K + extra;
        int threads = need <= 32 * 1024 ? 256 : (need <= 76 * 1024 ? 512 : 1024);
        while (threads > 128 && (long)ms3d_divup(ntiles, threads / 64) * ny < 128) threads >>= 1;  // few tiles: more blocks
        g.threads = threads;
        g.nblk = ms3d_divup(ntiles, threads / 64);
        if (g.nblk > 1024) g.nblk = 1024;
        g.lds = need;
    } else {
        g.G = (int)((LDS_BUDGET - extra) / per_offset);
        if (g.G > K) g.G = K;
        // the weight groups fill the LDS, so a CU holds one workgroup: as many rounds of 256 workgroups as 16-wave
        // workgroups need, then the waves per workgroup that spread the tiles evenly over those rounds (3136 tiles:
        // 242 x 13 waves instead of 196 x 16 with 60 CUs idle; 12544 tiles: 4 rounds of 13 waves instead of 3 + 1/16)
        static const int env_waves = [] { const char *e = getenv("MS3D_STREAM_WAVES"); return e ? atoi(e) : 0; }();
        const int rounds = ms3d_divup(ms3d_divup(ntiles, 16), 256);
        int waves = env_waves > 0 ? env_waves : ms3d_divup(ntiles, rounds * 256);
        waves = waves < 2 ? 2 : (waves > 16 ? 16 : waves);
        g.threads = waves * 64;
        g.nblk = ms3d_divup(ntiles, waves);
        g.lds = per_offset * g.G + extra;
    }
    if (g.nblk < 1) g.nblk = 1;
    g.ok = g.G >= 1 && g.nbt >= 1 && g.nbt <= MAX_NBT;
    return g;
}
}  // namespace

extern "C" {

// 1 if a layer of this shape may be served by the weight-streaming kernel (either direction), i.e. needs streamed images
int ms3d_spconv_wants_stream_image(int K, int Cin, int Cout)
{
    return (pairstream_layer_ok(K, Cin, Cout) || pairstream_layer_ok(K, Cout, Cin)) ? 1 : 0;
}

// what the aux slot behind a layer's weight images holds: 0 nothing, 1 the streamed f32 image (rectangular layers the
// weight-streaming kernel may serve), 2 the three-piece bf16 image (wide square layers; MS3D_BF16X3=0 switches it off)
int ms3d_spconv_aux_kind(int K, int Cin, int Cout)
{
    if (bf3_enabled() && bf3_dims_ok(K, Cin, Cout)) return 2;
    return ms3d_spconv_wants_stream_image(K, Cin, Cout) ? 1 : 0;
}

int ms3d_kmap_pairlist_wanted(int K, int Vout) { return pairlist_min_rows() >= 0 && Vout >= pairlist_min_rows() && K > 1 && K <= 27; }

// rows per tile of the pair list a forward / backward-data convolution of this shape walks: 0 = none (table walk /
// small-level kernel), 64 = ms3d_kmap_pairlist_build, 128 = ms3d_kmap_pairlist_build_rows(.., 128, ..)
int ms3d_spconv_pairlist_rows(int Vout, int K, int Cin, int Cout)
{
    const FwdGeom g = fwd_geometry(Vout, K, Cin, Cout, false, MS3D_PL_ROWS);
    return g.stream ? PSR : (g.pairlist ? g.pl_rows : 0);
}

int ms3d_spconv_pairlist_rows_dense(int Vout, int K, int Cin, int Cout)
{
    const int plain = ms3d_spconv_pairlist_rows(Vout, K, Cin, Cout);
    static const int narrow = env_int("MS3D_PL_NARROW", 1);    // 0: never build the 32-row lists
    if (plain != MS3D_PL_ROWS || !narrow || K != 27 || Cin != 32 || Cout != 32) return plain;
    const FwdGeom g = fwd_geometry(Vout, K, Cin, Cout, false, PL_ROWS_NARROW);
    return (g.ok && g.pairlist && g.pl_rows == PL_ROWS_NARROW) ? PL_ROWS_NARROW : plain;
}

int ms3d_spconv_partial_blocks(int Vout, int K, int Cin, int Cout, int with_pairlist)
{
    const FwdGeom g = fwd_geometry(Vout, K, Cin, Cout, true, with_pairlist == 1 ? MS3D_PL_ROWS : with_pairlist);
    return g.nblk * g.ny;
}

// out = conv(act(in)) [+ residual]; see ConvArgs.  wf from ms3d_spconv_prep_weights.
static int spconv_forward_impl(const float *in, const float *wf, const int *nbr, int Vout, int K, int Cin, int Cout,
                               float *out, const float *pre_scale, const float *pre_shift, int pre_relu,
                               const float *residual, const float *bn_x, const float *bn_scale, const float *bn_shift,
                               const float *bn_mean, const float *bn_invstd, float *bn_partial, int out_stats,
                               const float *bias, const int *pl_tile_start, const int *pl_entries, const float *wf_aux,
                               int aux_kind, ms3d_stream_t stream_);

int ms3d_spconv_forward(const float *in, const float *wf, const int *nbr, int Vout, int K, int Cin, int Cout,
                        float *out, const float *pre_scale, const float *pre_shift, int pre_relu,
                        const float *residual, const float *bn_x, const float *bn_scale, const float *bn_shift,
                        const float *bn_mean, const float *bn_invstd, float *bn_partial, int out_stats,
                        const float *bias, const int *pl_tile_start, const int *pl_entries, const float *wf_stream,
                        ms3d_stream_t stream_)
{
    return spconv_forward_impl(in, wf, nbr, Vout, K, Cin, Cout, out, pre_scale, pre_shift, pre_relu, residual, bn_x, bn_scale,
                               bn_shift, bn_mean, bn_invstd, bn_partial, out_stats, bias, pl_tile_start, pl_entries,
                               wf_stream, wf_stream ? 1 : 0, stream_);
}

static int spconv_forward_impl(const float *in, const float *wf, const int *nbr, int Vout, int K, int Cin, int Cout,
                               float *out, const float *pre_scale, const float *pre_shift, int pre_relu,
                               const float *residual, const float *bn_x, const float *bn_scale, const float *bn_shift,
                               const float *bn_mean, const float *bn_invstd, float *bn_partial, int out_stats,
                               const float *bias, const int *pl_tile_start, const int *pl_entries, const float *wf_stream,
                               int aux_kind, ms3d_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (Vout <= 0) return 0;
    ConvArgs p;
    p.bias = bias;
    p.wfs = nullptr;
    p.wfb = nullptr;
    if (aux_kind != 1 && aux_kind != 2) wf_stream = nullptr;
    p.pl_tile_start = pl_tile_start; p.pl_entries = pl_entries;
    p.out_stats = (out_stats && bn_partial && !bn_x) ? 1 : 0;
    static const int dyn_picks = [] { const char *e = getenv("MS3D_PL_DYNAMIC"); return e ? atoi(e) : 0; }();
    p.dyn_picks = dyn_picks;
    static const int k1_path = [] { const char *e = getenv("MS3D_K1_PATH"); return e ? atoi(e) : 3; }();
    p.k1_path = k1_path;
    p.in = in; p.wf = wf; p.nbr = nbr; p.out = out; p.pre_scale = pre_scale; p.pre_shift = pre_shift;
    p.residual = residual; p.bn_x = bn_x; p.bn_scale = bn_scale; p.bn_shift = bn_shift; p.bn_mean = bn_mean;
    p.bn_invstd = bn_invstd; p.bn_partial = bn_partial; p.Vout = Vout; p.K = K; p.Cin = Cin; p.Cout = Cout;
    p.NCH = ms3d_divup(Cin, 16); p.NBtot = ms3d_divup(Cout, 16); p.ntiles = ms3d_divup(Vout, 16); p.pre_relu = pre_relu;
    const FwdGeom g = fwd_geometry(Vout, K, Cin, Cout, bn_x != nullptr || p.out_stats,
                                   (pl_tile_start && pl_entries) ? ms3d_kmap_pairlist_rows_of(pl_tile_start) : 0);
    if (!g.ok) return MS3D_E_UNSUPPORTED;
    p.G = g.G;
    dim3 grid(g.nblk, g.ny);
    const bool aligned = (Cin % 16 == 0);
    p.ws_slabs = nullptr; p.ws_cnt = nullptr; p.ws_ng = p.ws_kg = p.ws_R = p.ws_tpp = 0;
    if (g.ws) {
        const int units = g.ws_R * g.ny;
        WsSpace *sp = ws_space(stream, g.ws_ng > 1 ? (size_t)g.ws_ng * p.ntiles * p.NBtot * 256 : 0, units);
        if (!sp) return (int)hipErrorOutOfMemory;
        p.ws_slabs = sp->slabs; p.ws_cnt = sp->cnt;
        p.ws_ng = g.ws_ng; p.ws_kg = g.ws_kg; p.ws_R = g.ws_R; p.ws_tpp = g.ws_tpp;
        const int nblk = ms3d_divup(units, 8) * 8 * g.ws_ng;
        switch (g.nbt) {
            case 1: return launch_fwd_ws<1>(p, nblk, g.threads, g.lds, stream);
            case 2: return launch_fwd_ws<2>(p, nblk, g.threads, g.lds, stream);
            case 3: return launch_fwd_ws<3>(p, nblk, g.threads, g.lds, stream);
            case 4: return launch_fwd_ws<4>(p, nblk, g.threads, g.lds, stream);
        }
        return MS3D_E_UNSUPPORTED;
    }
    if (g.stream) {
        if (!wf_stream || aux_kind != 1) return MS3D_E_UNSUPPORTED;  // the geometry (and the caller's partial buffer) assume this kernel
        p.wfs = wf_stream;
#define MS3D_PS(NBT_, CG_) \
    if (g.nbt == NBT_ && g.cg == CG_) return launch_fwd_pairstream<NBT_, CG_>(p, grid, g.threads, g.lds, stream);
        MS3D_PS(1, 1) MS3D_PS(1, 2) MS3D_PS(1, 3) MS3D_PS(1, 4)
        MS3D_PS(2, 1) MS3D_PS(2, 2) MS3D_PS(2, 3) MS3D_PS(2, 4)
        MS3D_PS(3, 1) MS3D_PS(3, 2) MS3D_PS(3, 3) MS3D_PS(3, 4)
        MS3D_PS(4, 1) MS3D_PS(4, 2) MS3D_PS(4, 3) MS3D_PS(4, 4)
#undef MS3D_PS
        return MS3D_E_UNSUPPORTED;
    }
    if (g.pairlist) {
        if (g.nbt == 1 && p.NCH == 1) return launch_fwd_pairlist<1, 1>(p, grid, g.threads, g.lds, stream);
        if (g.nbt == 1 && p.NCH == 2) return launch_fwd_pairlist<1, 2>(p, grid, g.threads, g.lds, stream);
        if (g.nbt == 2 && p.NCH == 1) return launch_fwd_pairlist<2, 1>(p, grid, g.threads, g.lds, stream);
        if (g.nbt == 2 && p.NCH == 2 && g.pl_rows == PL_ROWS_NARROW) return launch_fwd_pairlist<2, 2, PL_ROWS_NARROW>(p, grid, g.threads, g.lds, stream);
        if (g.nbt == 2 && p.NCH == 2) return launch_fwd_pairlist<2, 2>(p, grid, g.threads, g.lds, stream);
        if (g.nbt == 1 && p.NCH == 3) return launch_fwd_pairlist<1, 3>(p, grid, g.threads, g.lds, stream);
        if (g.nbt == 1 && p.NCH == 4) return launch_fwd_pairlist<1, 4>(p, grid, g.threads, g.lds, stream);
        return MS3D_E_UNSUPPORTED;
    }
    p.RT = g.small ? g.rt : 1;
    if (g.small && g.rt > 1 && aux_kind == 2 && wf_stream) {
        // the geometry (and the caller's statistics partials) count RT tiles per block
        p.wfb = wf_stream;
#define MS3D_RT_CASE(N) case N: return launch_fwd_small_bf3_rt<N, 3>(p, grid, g.threads, g.lds, stream);
        switch (g.nbt) {
            MS3D_RT_CASE(2) MS3D_RT_CASE(3) MS3D_RT_CASE(4) MS3D_RT_CASE(5) MS3D_RT_CASE(6) MS3D_RT_CASE(7) MS3D_RT_CASE(8)
        }
#undef MS3D_RT_CASE
        return MS3D_E_UNSUPPORTED;
    }
    if (g.small && aux_kind == 2 && wf_stream && bf3_dims_ok(K, Cin, Cout) && g.nbt >= 2) {
        p.wfb = wf_stream;
        switch (g.nbt) {
            case 2: return launch_fwd_small_bf3<2>(p, grid, g.threads, g.lds, stream);
            case 3: return launch_fwd_small_bf3<3>(p, grid, g.threads, g.lds, stream);
            case 4: return launch_fwd_small_bf3<4>(p, grid, g.threads, g.lds, stream);
            case 5: return launch_fwd_small_bf3<5>(p, grid, g.threads, g.lds, stream);
            case 6: return launch_fwd_small_bf3<6>(p, grid, g.threads, g.lds, stream);
            case 7: return launch_fwd_small_bf3<7>(p, grid, g.threads, g.lds, stream);
            case 8: return launch_fwd_small_bf3<8>(p, grid, g.threads, g.lds, stream);
        }
    }
    if (g.small) {
        switch (g.nbt) {
            case 1: return launch_fwd_small<1>(p, grid, g.threads, g.lds, aligned, stream);
            case 2: return launch_fwd_small<2>(p, grid, g.threads, g.lds, aligned, stream);
            case 3: return launch_fwd_small<3>(p, grid, g.threads, g.lds, aligned, stream);
            case 4: return launch_fwd_small<4>(p, grid, g.threads, g.lds, aligned, stream);
            case 5: return launch_fwd_small<5>(p, grid, g.threads, g.lds, aligned, stream);
            case 6: return launch_fwd_small<6>(p, grid, g.threads, g.lds, aligned, stream);
            case 7: return launch_fwd_small<7>(p, grid, g.threads, g.lds, aligned, stream);
            case 8: return launch_fwd_small<8>(p, grid, g.threads, g.lds, aligned, stream);
        }
        return MS3D_E_UNSUPPORTED;
    }
    if (aux_kind == 2 && wf_stream && bf3_dims_ok(K, Cin, Cout) && g.nbt >= 2) {
        // same grid as the f32 table walk (the statistics partials are sized for it); only the staged group shrinks:
        // three bf16 pieces are 6 bytes per weight
        const int nc32 = (Cin + 31) / 32;
        const size_t per_slab = (size_t)g.nbt * 3 * 1024;                 // one (offset, 32-channel chunk) of the slice
        const size_t extra = (bn_x != nullptr || p.out_stats) ? STAT_MAX_WAVES * stat_slot_floats(g.nbt) * sizeof(float) : 0;
        const int slabs = (int)((LDS_BUDGET - extra) / per_slab);
        // a stage = one gather round's offsets x as many chunks as fit beside them.  (Layers whose f32 image is LDS
        // resident keep the f32 kernel: its grid is the persistent one.)
        const int og = bf3_round_offsets(g.nbt);
        int go = K < og ? K : og;       // (2..5 offsets per stage measured within 3 % of each other)
        if (go > slabs) go = slabs;
        int gc = go >= 1 ? slabs / go : 0;
        if (gc > nc32) gc = nc32;
        if (g.G < K && gc >= 1) {
            p.wfb = wf_stream;
            p.G = go;
            p.GC = gc;
            const size_t lds = per_slab * p.G * p.GC + extra;
            switch (g.nbt) {
                case 2: return launch_fwd_bf3<2>(p, grid, g.threads, lds, stream);
                case 3: return launch_fwd_bf3<3>(p, grid, g.threads, lds, stream);
                case 4: return launch_fwd_bf3<4>(p, grid, g.threads, lds, stream);
                case 5: return launch_fwd_bf3<5>(p, grid, g.threads, lds, stream);
                case 6: return launch_fwd_bf3<6>(p, grid, g.threads, lds, stream);
                case 7: return launch_fwd_bf3<7>(p, grid, g.threads, lds, stream);
                case 8: return launch_fwd_bf3<8>(p, grid, g.threads, lds, stream);
            }
        }
    }
    switch (g.nbt) {
        case 1: return launch_fwd<1>(p, grid, g.threads, g.lds, aligned, stream);
        case 2: return launch_fwd<2>(p, grid, g.threads, g.lds, aligned, stream);
        case 3: return launch_fwd<3>(p, grid, g.threads, g.lds, aligned, stream);
        case 4: return launch_fwd<4>(p, grid, g.threads, g.lds, aligned, stream);
        case 5: return launch_fwd<5>(p, grid, g.threads, g.lds, aligned, stream);
        case 6: return launch_fwd<6>(p, grid, g.threads, g.lds, aligned, stream);
        case 7: return launch_fwd<7>(p, grid, g.threads, g.lds, aligned, stream);
        case 8: return launch_fwd<8>(p, grid, g.threads, g.lds, aligned, stream);
    }
    return MS3D_E_UNSUPPORTED;
}

// column blocks the offset-list backward-weight kernel serves: up to 4 (one slice); the K = 8 layers (strided / transposed
// convolutions between the levels) up to 14 in slices on grid.y -- their alternative is the f32 table walk, which visits
// every table entry of a 1-in-8 table (128 -> 96 at 51k rows: 234 us, 0.03 of the roofline); the K = 27 layers beyond 64
// channels keep the bf16x3 kernel.  MS3D_WGRAD_LIST_K8=0: the table walk as before.
static bool wgrad_list_cols_ok(int K, int nb)
{
    static const bool k8 = [] { const char *e = getenv("MS3D_WGRAD_LIST_K8"); return !e || atoi(e) != 0; }();
    return nb <= 4 || (k8 && K == 8 && nb <= 14);
}

static bool wgrad_bf3_ok(int Vout, int K, int Cin, int Cout, bool use_list)
{
    // below ~30 M row x channel x channel products the operand pass and the coarser row chunks cost more than the shorter
    // MFMA sequence saves (measured: 80 ch @ 2.5k rows and 48 ch @ 11.7k rows lose, 128 @ 2.5k and 64 @ 11.7k win)
    if ((double)Vout * Cin * Cout < 30e6) return false;
    static const bool wg = [] { const char *e = getenv("MS3D_BF16X3_WGRAD"); return !e || atoi(e) != 0; }();
    const int nb = ms3d_divup(Cout, 16);
    // K == 27: submanifold tables, input rows = output rows (the exported entry point is not told the input row count; a
    // K = 27 table whose input has OTHER rows than its output -- not a submanifold map, nothing in this package builds
    // one -- must not be handed to ms3d_spconv_backward_weight at these sizes: documented in the header).
    // Cin <= 256: the packed gather entry keeps the source word in 8 bits (as bf3_dims_ok)
    return bf3_enabled() && wg && !use_list && K == 27 && nb >= 3 && nb <= 8 && Cin >= 48 && Cin <= 256 && Cin % 16 == 0 &&
           Cout % 16 == 0;
}
int ms3d_spconv_wgrad_is_bf16x3(int Vout, int K, int Cin, int Cout, int offset_list) { return wgrad_bf3_ok(Vout, K, Cin, Cout, offset_list != 0) ? 1 : 0; }
// 1 when a backward-weight call of this shape takes the f32 table walk -- the kernel ms3d_spconv_layer_backward can leave
// to a batched launch (offset_list: an offset list of the table is passed)
int ms3d_spconv_wgrad_is_table_walk(int Vout, int K, int Cin, int Cout, int offset_list)
{
    const bool use_list = offset_list && wgrad_list_cols_ok(K, ms3d_divup(Cout, 16)) && K <= 27 && Cin % 16 == 0 && Cout % 16 == 0;
    return (Vout > 0 && !use_list && !wgrad_bf3_ok(Vout, K, Cin, Cout, use_list) && ms3d_divup(Cout, 16) <= 14) ? 1 : 0;
}
// K = 1 with a small weight (the per-point Linear layers of the heads: 575k rows x 16 -> 16 / 20 / 3): the table walk's
// grid is (row chunks) x 1 x Cin / 16, so 256 chunks are ONE workgroup per CU walking 2200 rows each in dependent trips of 32
// (107 us per launch); the slabs are a few hundred floats, so the rows are cut 4x finer instead (measured below)
static int wgrad_k1_chunks(int Vout, int K, int Cin, int Cout)
{
    // us per call incl. the slab reduction, 575k rows x 16 -> 16 / 20 / 3, by chunk limit (tools/scripts/k1_sweep.sh):
    // 256: 103 / 96 / 100;  1024: 70 / 73 / 68;  2048: 76 / 81 / 73;  4096: 109 / 120 / 102
    static const int k1_max = [] { const char *e = getenv("MS3D_WGRAD_K1_CHUNKS"); return e ? atoi(e) : 1024; }();
    if (K != 1 || (long)Cin * Cout > 4096 || k1_max <= 0) return 0;
    int c = ms3d_divup(Vout, 128);
    return c > k1_max ? k1_max : c;
}
// slabs + (bf16x3 kernel) the dout operand image and the activated input pieces
size_t ms3d_spconv_wgrad_ws_floats(int Vout, int K, int Cin, int Cout)
{
    const int k1 = wgrad_k1_chunks(Vout, K, Cin, Cout), rc = ms3d_spconv_wgrad_row_chunks(Vout);
    size_t n = (size_t)(k1 > rc ? k1 : rc) * K * Cin * Cout + 64;
    if (wgrad_bf3_ok(Vout, K, Cin, Cout, false)) {
        n += (size_t)ms3d_divup(Vout, 32) * ms3d_divup(Cout, 16) * 3 * 64 * 4 + 8;        // dout image, 16 B units
        n += (size_t)Vout * Cin * 3 / 2 + 8;                                                 // 6 B per input element
    }
    return n;
}

int ms3d_spconv_wgrad_row_chunks(int Vout)
{
    // ~1024+ waves in flight at full resolution, at most 256 partial slabs (the slab reduction reads chunks * |dW|)
    static const int max_chunks = [] {
        const char *e = getenv("MS3D_WGRAD_MAX_CHUNKS");  // tuning knob
        return e ? atoi(e) : 256;
    }();
    int chunks = ms3d_divup(Vout, 256);
    if (chunks > max_chunks) chunks = max_chunks;
    return chunks < 1 ? 1 : chunks;
}

// defer_nblk != NULL: the slab reduction is NOT launched; *defer_nblk = number of slabs left in partial_ws (0: dW is
// already final), to be reduced later by ms3d_wgrad_reduce_multi
static int spconv_backward_weight_impl(const float *in, const float *dout, const int *nbr, int Vout, int K, int Cin, int Cout,
                                       float *dW, const float *pre_scale, const float *pre_shift, int pre_relu,
                                       float *partial_ws, const int *ol_kt_start, const int *ol_entries, int *defer_nblk,
                                       ms3d_stream_t stream_, void *defer_launch = nullptr);

int ms3d_spconv_backward_weight(const float *in, const float *dout, const int *nbr, int Vout, int K, int Cin, int Cout,
                                float *dW, const float *pre_scale, const float *pre_shift, int pre_relu,
                                float *partial_ws, const int *ol_kt_start, const int *ol_entries, ms3d_stream_t stream_)
{
    return spconv_backward_weight_impl(in, dout, nbr, Vout, K, Cin, Cout, dW, pre_scale, pre_shift, pre_relu, partial_ws,
                                       ol_kt_start, ol_entries, nullptr, stream_);
}

static int spconv_backward_weight_impl(const float *in, const float *dout, const int *nbr, int Vout, int K, int Cin, int Cout,
                                       float *dW, const float *pre_scale, const float *pre_shift, int pre_relu,
                                       float *partial_ws, const int *ol_kt_start, const int *ol_entries, int *defer_nblk,
                                       ms3d_stream_t stream_, void *defer_launch)
{
    hipStream_t stream = (hipStream_t)stream_;
    const long n = (long)K * Cin * Cout;
    if (defer_nblk) *defer_nblk = 0;
    // defer_launch (a host buffer of ms3d_spconv_wgrad_launch_bytes(), only together with defer_nblk): when this call
    // takes the f32 table walk, the kernel is NOT launched either -- the buffer describes it for ms3d_spconv_wgrad_multi;
    // variant 0 = launched here as usual
    WgradLaunch *dl = defer_nblk ? static_cast<WgradLaunch *>(defer_launch) : nullptr;
    if (dl) dl->variant = 0;
    if (Vout <= 0) {
        MS3D_CHECK(hipMemsetAsync(dW, 0, sizeof(float) * n, stream));
        return 0;
    }
    WgradArgs p;
    p.in = in; p.dout = dout; p.nbr = nbr; p.partial = partial_ws; p.pre_scale = pre_scale; p.pre_shift = pre_shift;
    p.Vout = Vout; p.K = K; p.Cin = Cin; p.Cout = Cout; p.NBtot = ms3d_divup(Cout, 16); p.pre_relu = pre_relu;
    int chunks = ms3d_spconv_wgrad_row_chunks(Vout);
    // wide layers: the slab reduction reads chunks x |dW| (64 -> 64: 256 x 442 KB = 113 MB per launch, as much as the
    // gathers); half the slabs still leave every CU several workgroups (K / KG x Cin / 16 of them per chunk)
    static const int wide_chunks = [] { const char *e = getenv("MS3D_WGRAD_WIDE_CHUNKS"); return e ? atoi(e) : 128; }();
    static const int xwide_chunks = [] { const char *e = getenv("MS3D_WGRAD_XWIDE_CHUNKS"); return e ? atoi(e) : 128; }();
    if (n >= 100000 && chunks > wide_chunks) chunks = wide_chunks;
    if (n >= 400000 && chunks > xwide_chunks) chunks = xwide_chunks;
    const bool use_list = ol_kt_start && ol_entries && wgrad_list_cols_ok(K, p.NBtot) && K <= 27 && Cin % 16 == 0 && Cout % 16 == 0;
    p.ol_kt_start = ol_kt_start; p.ol_entries = ol_entries;
    int nblk;
    // offset-list kernel at 48 / 64 input channels: 64 parts x 3-4 input chunks is one workgroup per CU, and half the slabs
    // (us per launch with 256 / 128 / 64 / 32 parts: 64 -> 64 at 196k rows 396 / 343 / 320 / 589, at 50k 143 / 141 / 122 /
    // 193, 48 -> 48 at 196k 232 / 231 / 215 / 394; the narrow layers want all 256: 32 -> 32 at 417k 124 / 190 / 321)
    static const int list_wide_chunks = [] { const char *e = getenv("MS3D_WGRAD_LIST_WIDE_CHUNKS"); return e ? atoi(e) : 64; }();
    // 64 output channels from 64+ input channels (K = 27; round 6): TWO input chunks per workgroup -- the dout rows (256 B per
    // pair) are fetched once per two chunks instead of once per chunk, 768 instead of 1280 gathered bytes per pair -- on 128
    // parts (the grid keeps 256 workgroups), one batch per trip (the second chunk's rows take the registers of the second
    // batch).  us per launch, one chunk x 64 parts | two chunks x 128 parts: 64 -> 64 at 196k rows 320-328 | 269-271, at 51k
    // rows 122.6 | 112; 48 -> 48 (three column blocks) does not gain (74.5 | 78 at 51k rows) and keeps one chunk.
    // MS3D_WGRAD_LIST_NCH2=0: one chunk as before.
    static const int nch2 = env_int("MS3D_WGRAD_LIST_NCH2", 1);
    const bool wide2 = nch2 && use_list && K == 27 && p.NBtot == 4 && Cin % 32 == 0 && Cin >= 64;
    if (use_list && ms3d_divup(Cin, 16) >= 3 && chunks > list_wide_chunks) chunks = wide2 ? 2 * list_wide_chunks : list_wide_chunks;
    if (use_list) {
        // workgroups = the list's MS3D_PL_PARTS equal-pair-count parts, merged in pairs until there are <= chunks
        int shift = 0;
        while ((MS3D_PL_PARTS >> shift) > chunks) shift++;
        p.rows_per_block = shift;  // the list kernel reads this field as the merge shift
        nblk = MS3D_PL_PARTS >> shift;
    } else {
        if (!wgrad_bf3_ok(Vout, K, Cin, Cout, false)) {
            // the f32 table walk: no more row chunks than fill the chip once (every chunk costs a slab of |dW|; measured:
            // 64 -> 96, K = 8, 50k rows 103 -> 86 us, 160 -> 160 at 2.5k rows 76 -> 69; everything else has fewer anyway)
            static const int rounds = [] { const char *e = getenv("MS3D_WGRAD_F32_ROUNDS"); return e ? atoi(e) : 1; }();
            const int nbq = p.NBtot;
            const int kg = nbq <= 3 ? (K >= 27 ? 9 : 8) : nbq <= 7 ? 4 : nbq == 8 ? 3 : 2;
            const int yz = ms3d_divup(K, kg) * ms3d_divup(Cin, 16);
            if (rounds > 0) {
                int c = (1024 * rounds) / yz;
                if (c < 4) c = 4;
                if (c < chunks) chunks = c;
            }
            const int k1 = wgrad_k1_chunks(Vout, K, Cin, Cout);
            if (k1 > chunks) chunks = k1;
        }
        p.rows_per_block = ms3d_divup(ms3d_divup(Vout, chunks), 16) * 16;
        nblk = ms3d_divup(Vout, p.rows_per_block);
    }
    const int nb = p.NBtot;
    if (nb > 14) return MS3D_E_UNSUPPORTED;
    int rc;
    if (use_list) {
        // two input chunks per workgroup when Cin allows (entries and dout rows fetched once for both)
        // more than four column blocks (K = 8 layers of the wide levels, round 5): two or more slices of <= 4 blocks on grid.y
        const int nbs = nb <= 4 ? nb : ms3d_divup(nb, ms3d_divup(nb, 4));
        const bool two = ms3d_divup(Cin, 16) % 2 == 0 && nbs <= 2;
        rc = nbs == 1 ? (two ? launch_wgrad_offsetlist<1, 2>(p, nblk, stream) : launch_wgrad_offsetlist<1, 1>(p, nblk, stream))
           : nbs == 2 ? (two ? launch_wgrad_offsetlist<2, 2>(p, nblk, stream) : launch_wgrad_offsetlist<2, 1>(p, nblk, stream))
           : nbs == 3 ? launch_wgrad_offsetlist<3, 1>(p, nblk, stream)
                      : (wide2 ? launch_wgrad_offsetlist<4, 2, 1>(p, nblk, stream) : launch_wgrad_offsetlist<4, 1>(p, nblk, stream));
        if (rc) return rc;
        if (defer_nblk) { *defer_nblk = nblk; return 0; }
        launch_wgrad_reduce(partial_ws, nblk, n, dW, stream);
        MS3D_LAUNCH_CHECK();
        return 0;
    }
    if (wgrad_bf3_ok(Vout, K, Cin, Cout, use_list)) {
        // wide submanifold layers: both operands pre-split into three bf16 pieces by two elementwise passes
        float *base = partial_ws + (size_t)ms3d_spconv_wgrad_row_chunks(Vout) * K * Cin * Cout;
        bf16x8 *img = reinterpret_cast<bf16x8 *>((reinterpret_cast<uintptr_t>(base) + 15) & ~(uintptr_t)15);
        const long ntile = ms3d_divup(Vout, 32);
        bf16x8 *xs = img + (size_t)ntile * nb * 3 * 64;
        const long t1 = ntile * nb * 64, t2 = (long)Vout * (Cin / 8);
        const int nb1 = (int)((t1 + 255) / 256), nb2 = (int)((t2 + 255) / 256);
        wgrad_bf3_operands_kernel<<<nb1 + nb2, 256, 0, stream>>>(dout, in, Vout, Cin, Cout, nb, ntile, nb1, pre_scale, pre_shift,
                                                                pre_relu, img, xs);
        MS3D_LAUNCH_CHECK();
        WgradBf3Args q;
        q.xs = xs; q.dout_img = img; q.nbr = nbr; q.partial = partial_ws; q.Vout = Vout; q.K = K; q.Cin = Cin; q.Cout = Cout;
        q.NBtot = nb;
        // Row chunks: two workgroups fit a CU (512 slots).  As many chunks as make the grid two whole rounds of them --
        // every slab is |dW| floats written and read again by the reduction (128 chunks of 128 -> 128: 450 MB, a third
        // of the launch), while a grid of 1.1 or 1.5 rounds idles half the chip in its last one.  Measured, us per launch
        // with 128 / 64 / 32 chunks: 128 -> 128 at 50k rows 367 / 327 / 282, 96 -> 96 at 196k rows 776 / 854 / 1075.
        {
            const int kg = nb <= 4 ? (nb == 3 ? 14 : 9) : 9;
            const int yz = ms3d_divup(K, kg) * ms3d_divup(Cin, 16);
            static const int rounds = [] { const char *e = getenv("MS3D_WGRAD_BF3_ROUNDS"); return e ? atoi(e) : 2; }();
            int c = (512 * rounds) / yz;
            if (c < 8) c = 8;
            if (c < chunks) chunks = c;
        }
        q.rows_per_block = ms3d_divup(ms3d_divup(Vout, chunks), 32) * 32;
        nblk = ms3d_divup(Vout, q.rows_per_block);
        // waves split the output columns (NW waves x NBW blocks of 16), a workgroup takes KG offsets
        // (KG offsets per workgroup, waves, 16-column blocks per wave): the largest tiles that stay inside 256 registers at
        // two workgroups per CU -- every larger one tried spills (profiles/r03_fwd_experiments.txt section 6)
        rc = nb == 3 ? launch_wgrad_bf3<14, 3, 1>(q, nblk, stream) : nb == 4 ? launch_wgrad_bf3<9, 4, 1>(q, nblk, stream)
           : nb <= 6 ? launch_wgrad_bf3<9, 3, 2>(q, nblk, stream) : launch_wgrad_bf3<9, 4, 2>(q, nblk, stream);
        if (rc) return rc;
        if (defer_nblk) { *defer_nblk = nblk; return 0; }
        launch_wgrad_reduce(partial_ws, nblk, n, dW, stream);
        MS3D_LAUNCH_CHECK();
        return 0;
    }
    // KG * NBT <= 28 accumulators of 4 VGPRs
    if (nb == 1) rc = (K >= 27) ? launch_wgrad<9, 1>(p, nblk, stream, dl) : launch_wgrad<8, 1>(p, nblk, stream, dl);
    else if (nb == 2) rc = (K >= 27) ? launch_wgrad<9, 2>(p, nblk, stream, dl) : launch_wgrad<8, 2>(p, nblk, stream, dl);
    else if (nb == 3) rc = (K >= 27) ? launch_wgrad<9, 3>(p, nblk, stream, dl) : launch_wgrad<8, 3>(p, nblk, stream, dl);
    else if (nb == 4) rc = launch_wgrad<4, 4>(p, nblk, stream, dl);
    else if (nb == 5) rc = launch_wgrad<4, 5>(p, nblk, stream, dl);
    else if (nb == 6) rc = launch_wgrad<4, 6>(p, nblk, stream, dl);
    else if (nb == 7) rc = launch_wgrad<4, 7>(p, nblk, stream, dl);
    else if (nb == 8) rc = launch_wgrad<3, 8>(p, nblk, stream, dl);
    else if (nb <= 10) rc = launch_wgrad<2, 10>(p, nblk, stream, dl);
    else if (nb <= 12) rc = launch_wgrad<2, 12>(p, nblk, stream, dl);
    else rc = launch_wgrad<2, 14>(p, nblk, stream, dl);
    if (rc) return rc;
    if (defer_nblk) { *defer_nblk = nblk; return 0; }
    launch_wgrad_reduce(partial_ws, nblk, n, dW, stream);
    MS3D_LAUNCH_CHECK();
    return 0;
}

// slab reductions of many layers in one launch.  descs (DEVICE memory): n_desc records of 32 bytes
// {const float *slabs; float *dW; int64 n; int32 nblk; int32 block_begin | geometry bit} in ascending block order,
// block counts from ms3d_wgrad_reduce_blocks (which also tells the geometry bit)
size_t ms3d_spconv_wgrad_launch_bytes(void) { return 128; }

// descs: DEVICE array of n_desc 128-byte launch descriptions of ONE variant (what ms3d_spconv_layer_backward wrote into
// wgrad_deferred_launch, the first int of each patched to the layer's first block), total_blocks = sum of gx*gy*gz
int ms3d_spconv_wgrad_multi(const void *descs, int n_desc, int total_blocks, int variant, ms3d_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (n_desc <= 0 || total_blocks <= 0) return 0;
    switch (variant) {
#define MS3D_WM(KG, NBT) case KG * 100 + NBT: return launch_wgrad_multi<KG, NBT>(descs, n_desc, total_blocks, stream);
        MS3D_WM(9, 1) MS3D_WM(8, 1) MS3D_WM(9, 2) MS3D_WM(8, 2) MS3D_WM(9, 3) MS3D_WM(8, 3) MS3D_WM(4, 4) MS3D_WM(4, 5)
        MS3D_WM(4, 6) MS3D_WM(4, 7) MS3D_WM(3, 8) MS3D_WM(2, 10) MS3D_WM(2, 12) MS3D_WM(2, 14)
#undef MS3D_WM
    }
    return MS3D_E_UNSUPPORTED;
}

int ms3d_wgrad_reduce_blocks(long n, const float *slabs, const float *dW, int *wide)
{
    const bool w = n >= 32768 && (n & 3) == 0 && ((uintptr_t)slabs & 15) == 0 && ((uintptr_t)dW & 15) == 0;
    if (wide) *wide = w ? 1 : 0;
    return w ? ms3d_divup(n / 4, 64) : ms3d_divup(n, 16);
}

int ms3d_wgrad_reduce_multi(const void *descs, int n_desc, int total_blocks, ms3d_stream_t stream)
{
    if (n_desc <= 0 || total_blocks <= 0) return 0;
    wgrad_reduce_multi_kernel<<<total_blocks, 256, 0, (hipStream_t)stream>>>(static_cast<const WgradReduceDesc *>(descs),
                                                                           n_desc);
    MS3D_LAUNCH_CHECK();
    return 0;
}

// training-mode batch statistics of x [V, C] -> mean, invstd, scale = gamma*invstd, shift = beta - mean*scale,
// running stats updated in place (momentum, unbiased variance) like torch.nn.BatchNorm1d
int ms3d_bn_stats(const float *x, long V, int C, float eps, float momentum, const float *gamma, const float *beta,
                  float *running_mean, float *running_var, float *mean, float *invstd, float *scale, float *shift,
                  float *partial_ws, int partial_rows, ms3d_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (V <= 0) return MS3D_E_UNSUPPORTED;
    int nblk = (int)((V + 255) / 256);
    if (nblk > partial_rows) nblk = partial_rows;
    if (nblk < 1) nblk = 1;
    const int rows_per_block = (int)((V + nblk - 1) / nblk);
    const int rc = launch_bn_partial_stats(x, V, C, partial_ws, nblk, rows_per_block, stream);
    if (rc) return rc;
    bn_finalize_stats_kernel<<<ms3d_divup(C, 16), 1024, 0, stream>>>(partial_ws, nblk, C, V, eps, momentum, gamma, beta,
                                                                  running_mean, running_var, mean, invstd, scale, shift);
    MS3D_LAUNCH_CHECK();
    return 0;
}

// out [C] = column sums of x [V, C] (the bias gradient of a per-point Linear layer over 10^5..10^6 rows: torch's sum(0)
// of such a tall matrix takes ~60 us).  Partials (sum, sum of squares) per block of rows, reduced in fixed order;
// partial_ws: partial_rows * 2 * C floats, out2c: 2 * C floats of which the first C are the result.
int ms3d_column_sum(const float *x, long V, int C, float *partial_ws, int partial_rows, float *out2c, ms3d_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (V <= 0) {
        MS3D_CHECK(hipMemsetAsync(out2c, 0, sizeof(float) * 2 * C, stream));
        return 0;
    }
    int nblk = (int)((V + 1023) / 1024);
    if (nblk > partial_rows) nblk = partial_rows;
    if (nblk < 1) nblk = 1;
    const int rows_per_block = (int)((V + nblk - 1) / nblk);
    const int rc = launch_bn_partial_stats(x, V, C, partial_ws, nblk, rows_per_block, stream);
    if (rc) return rc;
    return ms3d_reduce_partials(partial_ws, nblk, 2 * C, out2c, stream_);
}

int ms3d_bn_finalize(const float *partial, int nparts, long V, int C, float eps, float momentum, const float *gamma,
                     const float *beta, float *running_mean, float *running_var, float *mean, float *invstd,
                     float *scale, float *shift, ms3d_stream_t stream)
{
    if (V <= 0) return MS3D_E_UNSUPPORTED;
    bn_finalize_stats_kernel<<<ms3d_divup(C, 16), 1024, 0, (hipStream_t)stream>>>(partial, nparts, C, V, eps, momentum, gamma,
                                                                                beta, running_mean, running_var, mean,
                                                                                invstd, scale, shift);
    MS3D_LAUNCH_CHECK();
    return 0;
}

int ms3d_bn_apply(const float *x, long V, int C, const float *scale, const float *shift, int relu, float *y,
                  ms3d_stream_t stream)
{
    const long n = V * C;
    if (n <= 0) return 0;
    bn_apply_kernel<<<(int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096), 256, 0, (hipStream_t)stream>>>(x, n, C, scale,
                                                                                                             shift, relu, y);
    MS3D_LAUNCH_CHECK();
    return 0;
}

// s1s2 [2][C] = column sums of a [nparts][2][C] partial buffer (fixed order -> deterministic)
int ms3d_reduce_partials(const float *partial, int nparts, int n, float *out, ms3d_stream_t stream)
{
    reduce_partial_kernel<<<ms3d_divup(n, 16), 1024, 0, (hipStream_t)stream>>>(partial, nparts, n, out);
    MS3D_LAUNCH_CHECK();
    return 0;
}

// s1s2 [2][C] = column sums of partial [nparts][2][C], and (dx != NULL) dx = scale * (dz - s1/V - xhat * s2/V) [+ add], in
// ONE launch (bn_bwd_reduce_apply_kernel); bit-identical to ms3d_reduce_partials + ms3d_bn_bwd_apply_add
int ms3d_bn_bwd_reduce_apply(const float *partial, int nparts, const float *dz, const float *x, long V, int C,
                             const float *scale, const float *mean, const float *invstd, const float *add, float *dx,
                             float *s1s2, ms3d_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    // OFF by default: measured SLOWER than the two launches it replaces (profiles/r04_experiments.txt section 6: 15.1 us
    // per launch against 7.6 + 4.9 us, median PointGroup step 20.99 against 20.48 ms) -- the hand-over of the sums through
    // atomic words costs every workgroup a fabric round trip at its start, more than the saved launch.  Kept as a tested,
    // bit-identical option (MS3D_BN_BWD_FUSED=1: 1110 -> 1033 launches per PointGroup step).
    static const bool fused = [] { const char *e = getenv("MS3D_BN_BWD_FUSED"); return e && atoi(e) != 0; }();
    if (!fused || 2 * C > 1024 || V <= 0) {
        int rc = ms3d_reduce_partials(partial, nparts, 2 * C, s1s2, stream_);
        if (rc || !dx) return rc;
        return ms3d_bn_bwd_apply_add(dz, x, V, C, scale, mean, invstd, s1s2, add, dx, stream_);
    }
    BnBwdState *st = bn_bwd_state(stream);
    if (!st) return MS3D_E_INTERNAL;
    const long n = V * C;
    const long work = dx ? ((C & 3) == 0 ? n / 4 : n) : 0;
    const int nred = ms3d_divup(2 * C, 16);
    long blocks = (work + 4095) / 4096;          // ~4 items per thread
    if (blocks > 2048) blocks = 2048;
    if (blocks < nred) blocks = nred;
    bn_bwd_reduce_apply_kernel<<<(int)blocks, 1024, 0, stream>>>(partial, nparts, dz, x, n, C, V, scale, mean, invstd, add, dx,
                                                               s1s2, st);
    MS3D_LAUNCH_CHECK();
    return 0;
}

int ms3d_bn_bwd_apply_add(const float *dz, const float *x, long V, int C, const float *scale, const float *mean,
                          const float *invstd, const float *s1s2, const float *add, float *dx, ms3d_stream_t stream)
{
    const long n = V * C;
    if (n <= 0) return 0;
    const long work = (C & 3) == 0 ? n / 4 : n;
    bn_bwd_apply_kernel<<<(int)((work + 255) / 256 < 8192 ? (work + 255) / 256 : 8192), 256, 0, (hipStream_t)stream>>>(
        dz, x, n, C, V, scale, mean, invstd, s1s2, add, dx);
    MS3D_LAUNCH_CHECK();
    return 0;
}
int ms3d_bn_bwd_apply(const float *dz, const float *x, long V, int C, const float *scale, const float *mean,
                      const float *invstd, const float *s1s2, float *dx, ms3d_stream_t stream)
{
    return ms3d_bn_bwd_apply_add(dz, x, V, C, scale, mean, invstd, s1s2, nullptr, dx, stream);
}

int ms3d_bn_bwd_partial(const float *dy, const float *x, long V, int C, const float *scale, const float *shift,
                        const float *mean, const float *invstd, int relu, float *dz, float *partial_ws,
                        int partial_rows, int *nparts_out, ms3d_stream_t stream)
{
    int nblk = (int)((V + 1023) / 1024);
    if (nblk > partial_rows) nblk = partial_rows;
    if (nblk < 1) nblk = 1;
    const int rows_per_block = (int)((V + nblk - 1) / nblk);
    if ((C & 3) == 0 && C <= 1024)
        bn_bwd_partial_kernel<4><<<nblk, 256, 0, (hipStream_t)stream>>>(dy, x, V, C, scale, shift, mean, invstd, relu, dz,
                                                                        partial_ws, rows_per_block);
    else if (C <= 256)
        bn_bwd_partial_kernel<1><<<nblk, 256, 0, (hipStream_t)stream>>>(dy, x, V, C, scale, shift, mean, invstd, relu, dz,
                                                                        partial_ws, rows_per_block);
    else
        return MS3D_E_UNSUPPORTED;
    MS3D_LAUNCH_CHECK();
    *nparts_out = nblk;
    return 0;
}


// ---------------------------------------------------------------------------------------------------------
// One-call layer entry points: everything a fused [BatchNorm -> ReLU ->] convolution needs per direction is
// enqueued from native code (one host->library transition per layer instead of five).
// ---------------------------------------------------------------------------------------------------------
size_t ms3d_spconv_layer_ws_floats(int Vin, int Vout, int K, int Cin, int Cout)
{
    // weight images (both orientations) + epilogue partials (fwd: Cout wide, bwd: Cin wide) + wgrad slabs + s1s2
    const size_t wf = ms3d_spconv_wf_floats(K, Cin, Cout) + ms3d_spconv_wf_floats(K, Cout, Cin);
    const auto blocks = [](int V, int K_, int ci, int co) {
        const int a = ms3d_spconv_partial_blocks(V, K_, ci, co, 0), b = ms3d_spconv_partial_blocks(V, K_, ci, co, 1);
        const int c = ms3d_spconv_partial_blocks(V, K_, ci, co, ms3d_spconv_pairlist_rows_dense(V, K_, ci, co));
        return (size_t)(a > b ? (a > c ? a : c) : (b > c ? b : c));
    };
    const size_t pf = blocks(Vout, K, Cin, Cout) * 2 * Cout;
    const size_t pb = blocks(Vin, K, Cout, Cin) * 2 * Cin;
    const size_t wg = ms3d_spconv_wgrad_ws_floats(Vout, K, Cin, Cout);
    return wf + (pf > pb ? pf : pb) + wg + 2 * (size_t)Cin + 64;
}

// forward: weight images (kept in wf_buf for the backward pass) + conv (+ fused input BN/ReLU, residual, bias) +
// optional output statistics (stat_partial [ms3d_spconv_partial_blocks(Vout,K,Cin,Cout)][2][Cout])
int ms3d_spconv_layer_forward(const float *x, const float *W, const int *nbr_fwd, int Vout, int K, int Cin, int Cout,
                              int mirror_bwd, const float *pre_scale, const float *pre_shift, int pre_relu,
                              const float *residual, const float *bias, float *wf_buf, float *y, float *stat_partial,
                              const int *pl_tile_start, const int *pl_entries, void *ev_start, void *ev_stop,
                              ms3d_stream_t stream)
{
    const size_t nwf = ms3d_spconv_wf_floats(K, Cin, Cout);  // wf_buf = [wf | aux (2n) | wft | aux (2n)], 6n floats
    float *wf = wf_buf, *wft = wf_buf + 3 * nwf;
    const int aux_kind = ms3d_spconv_aux_kind(K, Cin, Cout);
    int rc = W ? prep_weights_pair_impl(W, K, Cin, Cout, mirror_bwd, wf, wft, wf + nwf, wft + nwf, aux_kind, (hipStream_t)stream) : 0;  // W == NULL: wf_buf is current
    if (rc) return rc;
    // optional HIP events bracketing ONLY the convolution kernel, on the stream it is launched on (bench.py roofline)
    if (ev_start) MS3D_CHECK(hipEventRecord((hipEvent_t)ev_start, (hipStream_t)stream));
    rc = spconv_forward_impl(x, wf, nbr_fwd, Vout, K, Cin, Cout, y, pre_scale, pre_shift, pre_relu, residual, nullptr,
                             nullptr, nullptr, nullptr, nullptr, stat_partial, stat_partial != nullptr, bias, pl_tile_start,
                             pl_entries, wf + nwf, aux_kind, stream);
    if (ev_stop) MS3D_CHECK(hipEventRecord((hipEvent_t)ev_stop, (hipStream_t)stream));
    return rc;
}

void *ms3d_event_create(void)
{
    hipEvent_t e = nullptr;
    return hipEventCreate(&e) == hipSuccess ? (void *)e : nullptr;
}
void ms3d_event_destroy(void *e) { if (e) (void)hipEventDestroy((hipEvent_t)e); }
int ms3d_event_record(void *e, ms3d_stream_t stream) { return (int)hipEventRecord((hipEvent_t)e, (hipStream_t)stream); }
// milliseconds between two recorded events (synchronises on `stop`); < 0 on error
float ms3d_event_elapsed_ms(void *start, void *stop)
{
    float ms = -1.f;
    if (hipEventSynchronize((hipEvent_t)stop) != hipSuccess) return -1.f;
    if (hipEventElapsedTime(&ms, (hipEvent_t)start, (hipEvent_t)stop) != hipSuccess) return -1.f;
    return ms;
}

// backward of the fused layer.  need_dx / bn (scale != NULL) select the pieces:
//   dz   = conv^T(dy) [* ReLU mask of the fused BN]           -> written to dx (in place: dx doubles as dz)
//   s1s2 = (sum dz, sum dz*xhat)  = (dbeta, dgamma)            -> dgb [2][Cin]
//   dx   = scale * (dz - s1/V - xhat*s2/V)  (training)  or  scale * dz  (eval)
//   dW   = sum_i act(x[nbr])^T dy  through per-chunk slabs in `ws`
// events that order the two streams of ms3d_spconv_layer_backward (no timing; re-recorded round robin: a wait that was
// already issued keeps the state it captured)
static hipEvent_t order_event()
{
    // one pool per device (an event belongs to the device that was current when it was created), creation under a lock
    // (a process-wide pool with a plain ready flag was wrong for several devices / threads per process)
    constexpr int MAX_DEV = 16, POOL = 256;
    static hipEvent_t pool[MAX_DEV][POOL];
    static bool ready[MAX_DEV][POOL];
    static unsigned next[MAX_DEV];
    static std::mutex lock;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return nullptr;
    std::lock_guard<std::mutex> guard(lock);
    const unsigned i = next[dev]++ & (POOL - 1);
    if (!ready[dev][i]) {
        if (hipEventCreateWithFlags(&pool[dev][i], hipEventDisableTiming) != hipSuccess) return nullptr;
        ready[dev][i] = true;
    }
    return pool[dev][i];
}

int ms3d_spconv_layer_backward(const float *x, const float *dy, const float *wf_buf, const int *nbr_fwd,
                               const int *nbr_bwd, int Vin, int Vout, int K, int Cin, int Cout, const float *scale,
                               const float *shift, const float *mean, const float *invstd, int pre_relu, int training,
                               int need_dx, float *dx, const float *dx_add, float *dgb, float *dW, float *ws,
                               const int *ol_fwd_kt_start, const int *ol_fwd_entries, const int *pl_bwd_tile_start,
                               const int *pl_bwd_entries, void *ev_start, void *ev_stop, void *ev_wg_start, void *ev_wg_stop,
                               float *ws_wgrad, ms3d_stream_t wgrad_stream, int join, float *wgrad_slabs,
                               int *wgrad_deferred_nblk, void *wgrad_deferred_launch, ms3d_stream_t stream)
{
    const size_t nwf = ms3d_spconv_wf_floats(K, Cin, Cout);
    const float *wft = wf_buf + 3 * nwf, *wfts = wf_buf + 4 * nwf;
    const int aux_kind = ms3d_spconv_aux_kind(K, Cin, Cout);
    const bool bn = scale != nullptr;
    int rc;
    // Backward-weight needs x and dy only, not the backward-data result: with a second stream (and its own slab
    // workspace) it runs BESIDE the backward-data convolution and the three small kernels of the BatchNorm-backward chain
    // behind it -- ~250 launches of ~5 us per step that otherwise leave the chip idle between two convolutions.
    hipStream_t main = (hipStream_t)stream, side = wgrad_stream ? (hipStream_t)wgrad_stream : main;
    float *slabs;
    hipEvent_t done = nullptr;
    auto run_wgrad = [&]() -> int {
        if (ev_wg_start) MS3D_CHECK(hipEventRecord((hipEvent_t)ev_wg_start, side));
        // wgrad_slabs: a slab area of the caller's that outlives this call; with wgrad_deferred_nblk the slab reduction is
        // left to one ms3d_wgrad_reduce_multi launch over many layers
        int r = spconv_backward_weight_impl(x, dy, nbr_fwd, Vout, K, Cin, Cout, dW, scale, shift, pre_relu,
                                            wgrad_slabs ? wgrad_slabs : slabs, ol_fwd_kt_start, ol_fwd_entries,
                                            wgrad_slabs ? wgrad_deferred_nblk : nullptr, (ms3d_stream_t)side,
                                            (wgrad_slabs && !ev_wg_start) ? wgrad_deferred_launch : nullptr);
        if (ev_wg_stop) MS3D_CHECK(hipEventRecord((hipEvent_t)ev_wg_stop, side));
        return r;
    };
    if (side != main) {
        if (!ws_wgrad) return MS3D_E_WORKSPACE;
        slabs = ws_wgrad;
        hipEvent_t ready = order_event();
        done = order_event();
        if (!ready || !done) return MS3D_E_INTERNAL;
        MS3D_CHECK(hipEventRecord(ready, main));          // dy (and the lists built on this stream) are complete here
        MS3D_CHECK(hipStreamWaitEvent(side, ready, 0));
        rc = run_wgrad();
        if (rc) return rc;
        MS3D_CHECK(hipEventRecord(done, side));
    }
    // optional HIP events bracketing ONLY the backward-data convolution kernel (bench.py roofline)
    if (ev_start && (need_dx || bn)) MS3D_CHECK(hipEventRecord((hipEvent_t)ev_start, main));
    if (need_dx || bn) {
        if (!bn) {
            // dx_add (the gradient that reaches x over a skip connection) rides in the residual epilogue
            rc = spconv_forward_impl(dy, wft, nbr_bwd, Vin, K, Cout, Cin, dx, nullptr, nullptr, 0, dx_add, nullptr, nullptr,
                                     nullptr, nullptr, nullptr, nullptr, 0, nullptr, pl_bwd_tile_start, pl_bwd_entries, wfts,
                                     aux_kind, stream);
            if (rc) return rc;
            if (ev_stop) MS3D_CHECK(hipEventRecord((hipEvent_t)ev_stop, main));
            ev_stop = nullptr;
        } else {
            if (!pre_relu) return MS3D_E_UNSUPPORTED;  // BN without ReLU in front of a conv: handled by the generic path
            const int nparts = ms3d_spconv_partial_blocks(Vin, K, Cout, Cin, (pl_bwd_tile_start && pl_bwd_entries)
                                                                                 ? ms3d_kmap_pairlist_rows_of(pl_bwd_tile_start) : 0);
            float *partial = ws;
            rc = spconv_forward_impl(dy, wft, nbr_bwd, Vin, K, Cout, Cin, dx, nullptr, nullptr, 0, nullptr, x, scale, shift,
                                     mean, invstd, partial, 0, nullptr, pl_bwd_tile_start, pl_bwd_entries, wfts, aux_kind,
                                     stream);
            if (rc) return rc;
            if (ev_stop) MS3D_CHECK(hipEventRecord((hipEvent_t)ev_stop, main));
            ev_stop = nullptr;
            if (need_dx && training) {
                // slab of sums + the elementwise BatchNorm-backward pass in one launch
                rc = ms3d_bn_bwd_reduce_apply(partial, nparts, dx, x, Vin, Cin, scale, mean, invstd, dx_add, dx, dgb, stream);
                if (rc) return rc;
            } else {
                rc = ms3d_reduce_partials(partial, nparts, 2 * Cin, dgb, stream);
                if (rc) return rc;
            }
            if (need_dx && !training) {
                {
                    if (dx_add) return MS3D_E_UNSUPPORTED;   // eval-mode statistics: the caller adds the skip gradient itself
                    rc = ms3d_bn_apply(dx, Vin, Cin, scale, nullptr, 0, dx, stream);  // dx = dz * scale
                }
                if (rc) return rc;
            }
        }
    }
    if (side != main) {
        // join = 1: everything queued on `stream` after this call sees dW (what a gradient hook of a data-parallel
        // wrapper needs); join = 0: the caller joins the two streams itself before dW is read
        if (join) MS3D_CHECK(hipStreamWaitEvent(main, done, 0));
        return 0;
    }
    const int pb0 = ms3d_spconv_partial_blocks(Vin, K, Cout, Cin, 0), pb1 = ms3d_spconv_partial_blocks(Vin, K, Cout, Cin, 1);
    const int pb2 = ms3d_spconv_partial_blocks(Vin, K, Cout, Cin, ms3d_spconv_pairlist_rows_dense(Vin, K, Cout, Cin));
    slabs = ws + (size_t)(pb0 > pb1 ? (pb0 > pb2 ? pb0 : pb2) : (pb1 > pb2 ? pb1 : pb2)) * 2 * Cin;
    return run_wgrad();
}

}  // extern "C"
