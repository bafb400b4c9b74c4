// HAIS hierarchical aggregation on the device.  Replaces hierarchical_aggregation
// (reference hierarchical_aggregation/hierarchical_aggregation.cpp:8-184: serial host BFS + split, and
// hierarchical_aggregation.cu:20-204: 11 cudaMalloc, two tiny kernels, a per-primary cudaMemcpy loop) and the
// tensor merge of functions/hais_ops.py:55-73.
//
//   1. connected components with the order-exact BFS core (bfs_cluster.hip, threshold 0 = keep every component);
//   2. per component: centre = SERIAL f32 sum of the shifted coordinates in BFS order / size (bit-identical to
//      find_cc's accum_x/y/z), class of the seed, scene of the seed, fragment / kept / primary by the
//      class-relative size thresholds 0.05*avg and 0.3*avg (double product rounded to f32, .cpp:58-59);
//   3. set aggregation: every fragment looks for the nearest primary of its class and scene and is absorbed if
//      d^2 < max(0.01*sqrt(n_primary), radius_avg[cls])^2 (.cu:27-63); absorbed fragments are appended in ascending
//      fragment index (canonical; the reference order is atomic-dependent), caps 1024 fragments / 8192 points;
//   4. output = kept fragments, then primaries (+ absorbed points), as (cluster_id, point) rows + offsets.
#include "common.h"
#include "scan.h"
#include "../../include/minsu3d_hip.h"

int ms3d_bfs_run_internal(int mode, int thr_i, float thr_f, int capped_hint, const int16_t *sem, const int *ball_idx,
                          long n_edges, const int *start_len, int N, int *cluster_idxs, int *cluster_offsets, int *counts,
                          void *workspace, size_t workspace_bytes, hipStream_t stream);

namespace {

constexpr int MAX_FRAG = 1024, MAX_PTS = 8192;

struct CC {  // per connected component
    float cx, cy, cz;
    int cls, batch, size, kind;  // kind: bit0 fragment, bit1 kept, bit2 primary
};

__global__ void ha_describe_kernel(int ncc, const int *__restrict__ cc_idx, const int *__restrict__ cc_off,
                                   const int16_t *__restrict__ sem, const float *__restrict__ coord_shift,
                                   const uint8_t *__restrict__ batch_idxs, const float *__restrict__ point_num_avg,
                                   CC *cc)
{
    // one WAVE per component: the lanes fetch 64 members at a time (index -> coordinates, two dependent loads that one
    // thread per component paid once per member: 2.4 ms on a 5000-point component), then every lane adds them up in
    // member order -- serial, in BFS order: the same float sum as the reference
    const int c = (int)((blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6);
    if (c >= ncc) return;
    const int l = lane_id();
    const int s = cc_off[c], e = cc_off[c + 1];
    float ax = 0.f, ay = 0.f, az = 0.f;
    for (int q0 = s; q0 < e; q0 += 64) {
        const int q = q0 + l;
        float x = 0.f, y = 0.f, z = 0.f;
        if (q < e) {
            const int i = cc_idx[q * 2 + 1];
            x = coord_shift[i * 3 + 0];
            y = coord_shift[i * 3 + 1];
            z = coord_shift[i * 3 + 2];
        }
        const int n = min(64, e - q0);
        if (n == 64) {
            // full batch: constant lane numbers (v_readlane with an immediate), no loop control between the dependent adds
#pragma unroll
            for (int k = 0; k < 64; k++) {
                ax += __shfl(x, k, 64);
                ay += __shfl(y, k, 64);
                az += __shfl(z, k, 64);
            }
        } else {
            for (int k = 0; k < n; k++) {
                ax += __shfl(x, k, 64);
                ay += __shfl(y, k, 64);
                az += __shfl(z, k, 64);
            }
        }
    }
    if (l != 0) return;
    const int seed = cc_idx[s * 2 + 1];
    const int size = e - s;
    CC o;
    o.size = size;
    o.cls = (int)sem[seed];
    o.batch = (int)batch_idxs[seed];
    o.cx = ax / (float)size;
    o.cy = ay / (float)size;
    o.cz = az / (float)size;
    const float mean = point_num_avg[o.cls];
    const float low = (float)(0.05 * (double)mean), high = (float)(0.3 * (double)mean);
    int kind = 0;
    if ((float)size < high) {
        kind |= 1;
        if ((float)size >= low) kind |= 2;
    } else
        kind |= 4;
    o.kind = kind;
    cc[c] = o;
}

__global__ void ha_nearest_kernel(int ncc, const CC *__restrict__ cc, const float *__restrict__ radius_avg, int *absorb_to)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= ncc) return;
    absorb_to[f] = -1;
    const CC me = cc[f];
    if (!(me.kind & 1)) return;
    float nearest = 10000.f;  // INFINITY_DIS_SQUARE
    int ni = -1;
    for (int p = 0; p < ncc; p++) {
        const CC o = cc[p];
        if (!(o.kind & 4) || o.cls != me.cls || o.batch != me.batch) continue;
        const float dx = o.cx - me.cx, dy = o.cy - me.cy, dz = o.cz - me.cz;
        const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        if (d < nearest) {
            nearest = d;
            ni = p;
        }
    }
    if (ni < 0) return;
    const float r_size = (float)(0.01 * (double)sqrtf((float)cc[ni].size));
    const float r_cls = radius_avg[me.cls];
    const float r_set = r_size > r_cls ? r_size : r_cls;
    if (nearest < __fmul_rn(r_set, r_set)) absorb_to[f] = ni;
}

// per component: rows it contributes to the output as a kept fragment / as a primary (+ absorbed points, capped)
__global__ void ha_sizes_kernel(int ncc, int using_set_aggr, const CC *__restrict__ cc, const int *__restrict__ absorb_to,
                                int *kept_flag, int *kept_rows, int *prim_flag, int *prim_rows)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncc) return;
    const int kind = cc[c].kind;
    kept_flag[c] = (kind & 2) ? 1 : 0;
    kept_rows[c] = (kind & 2) ? cc[c].size : 0;
    int rows = 0;
    if (kind & 4) {
        rows = cc[c].size;
        if (using_set_aggr) {
            int nfrag = 0, npts = 0;
            for (int f = 0; f < ncc && nfrag < MAX_FRAG; f++) {
                if (absorb_to[f] != c) continue;
                nfrag++;
                npts += min(cc[f].size, MAX_PTS - npts);
            }
            rows += npts;
        }
    }
    prim_flag[c] = (kind & 4) ? 1 : 0;
    prim_rows[c] = rows;
}

// totals: [0] n_kept  [1] kept rows  [2] n_primary  [3] primary rows
__global__ void ha_emit_kernel(int ncc, int using_set_aggr, const CC *__restrict__ cc, const int *__restrict__ cc_idx,
                               const int *__restrict__ cc_off, const int *__restrict__ absorb_to,
                               const int *__restrict__ kept_id, const int *__restrict__ kept_off,
                               const int *__restrict__ prim_id, const int *__restrict__ prim_off,
                               const int *__restrict__ totals, int *out_idx, int *out_off)
{
    const int c = blockIdx.x;  // one block per component
    if (c >= ncc) return;
    const int kind = cc[c].kind;
    const int s = cc_off[c], size = cc[c].size;
    const int n_kept = totals[0], kept_rows_total = totals[1];
    if (kind & 2) {
        const int id = kept_id[c], base = kept_off[c];
        for (int q = threadIdx.x; q < size; q += blockDim.x) {
            out_idx[(base + q) * 2 + 0] = id;
            out_idx[(base + q) * 2 + 1] = cc_idx[(s + q) * 2 + 1];
        }
        if (threadIdx.x == 0) out_off[id] = base;
    }
    if (kind & 4) {
        const int id = n_kept + prim_id[c], base = kept_rows_total + prim_off[c];
        for (int q = threadIdx.x; q < size; q += blockDim.x) {
            out_idx[(base + q) * 2 + 0] = id;
            out_idx[(base + q) * 2 + 1] = cc_idx[(s + q) * 2 + 1];
        }
        if (threadIdx.x == 0) {
            out_off[id] = base;
            if (using_set_aggr) {  // absorbed fragments: ascending fragment index, serial caps as concat_fragments_
                int nfrag = 0, npts = 0, w = base + size;
                for (int f = 0; f < ncc && nfrag < MAX_FRAG; f++) {
                    if (absorb_to[f] != c) continue;
                    nfrag++;
                    for (int q = cc_off[f]; q < cc_off[f + 1] && npts < MAX_PTS; q++, npts++, w++) {
                        out_idx[w * 2 + 0] = id;
                        out_idx[w * 2 + 1] = cc_idx[q * 2 + 1];
                    }
                }
            }
        }
    }
    if (c == 0 && threadIdx.x == 0) out_off[n_kept + totals[2]] = kept_rows_total + totals[3];
}


// ---- the reference's own output contract (hierarchical_aggregation.cpp:105-184): the four lists separately ----
// flags/ids per component: fragment (kind & 1), kept (kind & 2), primary (kind & 4); one block per component
__global__ void ha_parts_kernel(int ncc, int using_set_aggr, const CC *__restrict__ cc, const int *__restrict__ cc_idx,
                                const int *__restrict__ cc_off, const int *__restrict__ absorb_to,
                                const int *__restrict__ kept_id, const int *__restrict__ kept_off,
                                const int *__restrict__ prim_id, const int *__restrict__ prim_off /* rows incl. absorbed */,
                                const int *__restrict__ frag_id, const int *__restrict__ frag_off,
                                const int *__restrict__ praw_off /* primary rows without absorbed */,
                                int *kept_idxs, int *kept_offsets, float *kept_centers, int *prim_idxs, int *prim_offsets,
                                float *prim_centers, int *frag_idxs, int *frag_offsets, float *frag_centers, int *post_idxs,
                                int *post_offsets)
{
    const int c = blockIdx.x;
    if (c >= ncc) return;
    const CC me = cc[c];
    const int s = cc_off[c], size = me.size;
    auto put = [&](int *idxs, int *offsets, float *centers, int id, int base) {
        for (int q = threadIdx.x; q < size; q += blockDim.x) {
            idxs[(base + q) * 2 + 0] = id;
            idxs[(base + q) * 2 + 1] = cc_idx[(s + q) * 2 + 1];
        }
        if (threadIdx.x == 0) {
            offsets[id + 1] = base + size;  // offsets[0] = 0 is written by the launcher's memset
            centers[id * 5 + 0] = me.cx;
            centers[id * 5 + 1] = me.cy;
            centers[id * 5 + 2] = me.cz;
            centers[id * 5 + 3] = (float)me.cls;
            centers[id * 5 + 4] = (float)me.batch;
        }
    };
    if (me.kind & 2) put(kept_idxs, kept_offsets, kept_centers, kept_id[c], kept_off[c]);
    if (me.kind & 4) put(prim_idxs, prim_offsets, prim_centers, prim_id[c], praw_off[c]);
    if (using_set_aggr && (me.kind & 1)) put(frag_idxs, frag_offsets, frag_centers, frag_id[c], frag_off[c]);
    if (using_set_aggr && (me.kind & 4)) {
        const int id = prim_id[c], base = prim_off[c];
        for (int q = threadIdx.x; q < size; q += blockDim.x) {
            post_idxs[(base + q) * 2 + 0] = id;
            post_idxs[(base + q) * 2 + 1] = cc_idx[(s + q) * 2 + 1];
        }
        if (threadIdx.x == 0) {
            int nfrag = 0, npts = 0, w = base + size;
            for (int f = 0; f < ncc && nfrag < MAX_FRAG; f++) {
                if (absorb_to[f] != c) continue;
                nfrag++;
                for (int q = cc_off[f]; q < cc_off[f + 1] && npts < MAX_PTS; q++, npts++, w++) {
                    post_idxs[w * 2 + 0] = id;
                    post_idxs[w * 2 + 1] = cc_idx[q * 2 + 1];
                }
            }
            post_offsets[id + 1] = w;
        }
    }
}

__global__ void ha_frag_sizes_kernel(int ncc, const CC *__restrict__ cc, int *frag_flag, int *frag_rows, int *praw_rows)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncc) return;
    const int kind = cc[c].kind;
    frag_flag[c] = (kind & 1) ? 1 : 0;
    frag_rows[c] = (kind & 1) ? cc[c].size : 0;
    praw_rows[c] = (kind & 4) ? cc[c].size : 0;
}

struct HaWs {
    int *cc_idx, *cc_off, *absorb_to, *kept_flag, *kept_rows, *prim_flag, *prim_rows, *kept_id, *kept_off, *prim_id,
        *prim_off, *totals, *frag_flag, *frag_rows, *frag_id, *frag_off, *praw_rows, *praw_off;
    CC *cc;
    float *avg;
    void *scan_ws, *bfs_ws;
    size_t bfs_bytes;
};
size_t carve(HaWs &w, int N, int nclass, void *base)
{
    size_t off = 0;
    auto take = [&](size_t bytes) {
        void *r = base ? (void *)((char *)base + off) : nullptr;
        off += ms3d_align(bytes);
        return r;
    };
    const size_t nb = sizeof(int) * (size_t)N;
    w.cc_idx = (int *)take(nb * 2); w.cc_off = (int *)take(nb + 4);
    w.absorb_to = (int *)take(nb); w.kept_flag = (int *)take(nb); w.kept_rows = (int *)take(nb);
    w.prim_flag = (int *)take(nb); w.prim_rows = (int *)take(nb); w.kept_id = (int *)take(nb);
    w.kept_off = (int *)take(nb); w.prim_id = (int *)take(nb); w.prim_off = (int *)take(nb);
    w.totals = (int *)take(sizeof(int) * 8);
    w.frag_flag = (int *)take(nb); w.frag_rows = (int *)take(nb); w.frag_id = (int *)take(nb); w.frag_off = (int *)take(nb);
    w.praw_rows = (int *)take(nb); w.praw_off = (int *)take(nb);
    w.cc = (CC *)take(sizeof(CC) * (size_t)N);
    w.avg = (float *)take(sizeof(float) * 2 * (size_t)(nclass > 0 ? nclass : 1));
    w.scan_ws = take(ms3d_scan_workspace_bytes());
    w.bfs_bytes = ms3d_bfs_workspace_bytes(N);
    w.bfs_ws = take(w.bfs_bytes);
    return off;
}

}  // namespace

extern "C" {

size_t ms3d_hais_workspace_bytes(int N, int nclass)
{
    HaWs w;
    return carve(w, N > 0 ? N : 1, nclass, nullptr);
}

int ms3d_hierarchical_aggregation(const int16_t *semantic_label, const float *coord_shift, const uint8_t *batch_idxs,
                                  const int *ball_query_idxs, long n_edges, const int *start_len, int N, int capped_hint,
                                  int using_set_aggr, const float *point_num_avg /*[host]*/,
                                  const float *radius_avg /*[host]*/, int nclass, int *cluster_idxs /*[2N,2]*/,
                                  int *cluster_offsets /*[N+1]*/, int *counts /*[host,2]*/, void *workspace,
                                  size_t workspace_bytes, ms3d_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    counts[0] = counts[1] = 0;
    if (N <= 0) {
        MS3D_CHECK(hipMemsetAsync(cluster_offsets, 0, sizeof(int), stream));
        return 0;
    }
    HaWs w;
    if (carve(w, N, nclass, workspace) > workspace_bytes) return MS3D_E_WORKSPACE;
    MS3D_CHECK(hipMemcpyAsync(w.avg, point_num_avg, sizeof(float) * nclass, hipMemcpyHostToDevice, stream));
    MS3D_CHECK(hipMemcpyAsync(w.avg + nclass, radius_avg, sizeof(float) * nclass, hipMemcpyHostToDevice, stream));
    int cc_counts[2];
    int rc = ms3d_bfs_run_internal(0, 0, 0.f, capped_hint, semantic_label, ball_query_idxs, n_edges, start_len, N, w.cc_idx, w.cc_off,
                                   cc_counts, w.bfs_ws, w.bfs_bytes, stream);
    if (rc) return rc;
    const int ncc = cc_counts[0];
    const int nb = ms3d_divup(ncc, 128);
    ha_describe_kernel<<<ms3d_divup((long)ncc * 64, 256), 256, 0, stream>>>(ncc, w.cc_idx, w.cc_off, semantic_label, coord_shift, batch_idxs, w.avg, w.cc);
    MS3D_LAUNCH_CHECK();
    if (using_set_aggr) {
        ha_nearest_kernel<<<nb, 128, 0, stream>>>(ncc, w.cc, w.avg + nclass, w.absorb_to);
        MS3D_LAUNCH_CHECK();
    }
    ha_sizes_kernel<<<nb, 128, 0, stream>>>(ncc, using_set_aggr, w.cc, w.absorb_to, w.kept_flag, w.kept_rows, w.prim_flag,
                                           w.prim_rows);
    MS3D_LAUNCH_CHECK();
    if ((rc = ms3d_exclusive_scan_i32(w.kept_flag, w.kept_id, ncc, w.totals + 0, w.scan_ws, stream))) return rc;
    if ((rc = ms3d_exclusive_scan_i32(w.kept_rows, w.kept_off, ncc, w.totals + 1, w.scan_ws, stream))) return rc;
    if ((rc = ms3d_exclusive_scan_i32(w.prim_flag, w.prim_id, ncc, w.totals + 2, w.scan_ws, stream))) return rc;
    if ((rc = ms3d_exclusive_scan_i32(w.prim_rows, w.prim_off, ncc, w.totals + 3, w.scan_ws, stream))) return rc;
    ha_emit_kernel<<<ncc, 128, 0, stream>>>(ncc, using_set_aggr, w.cc, w.cc_idx, w.cc_off, w.absorb_to, w.kept_id, w.kept_off,
                                           w.prim_id, w.prim_off, w.totals, cluster_idxs, cluster_offsets);
    MS3D_LAUNCH_CHECK();
    int h[4];
    MS3D_CHECK(hipMemcpyAsync(h, w.totals, sizeof(int) * 4, hipMemcpyDeviceToHost, stream));
    MS3D_CHECK(hipStreamSynchronize(stream));
    counts[0] = h[0] + h[2];
    counts[1] = h[1] + h[3];
    return 0;
}

// The reference's own output contract (hierarchical_aggregation.h:14-28, .cpp:105-184): kept fragments, primaries,
// and -- with set aggregation -- all fragments and the primaries with their absorbed fragments, each as
// (idxs [rows,2], offsets [n+1], centers [n,5] = x, y, z, class, scene).  Every buffer has capacity N rows /
// N+1 offsets / N*5 floats; post_idxs rows beyond post_offsets[n_primary] are zero (the reference zero-fills
// sumNPoint_fragment + sumNPoint_primary rows and the wrapper cuts the tail, functions/hais_ops.py:60-63).
// counts [host,8]: n_kept, kept rows, n_primary, primary rows (post, incl. absorbed), n_fragment, fragment rows,
// primary rows (raw), 0.
int ms3d_hierarchical_aggregation_parts(const int16_t *semantic_label, const float *coord_shift, const uint8_t *batch_idxs,
                                        const int *ball_query_idxs, long n_edges, const int *start_len, int N,
                                        int capped_hint, int using_set_aggr, const float *point_num_avg /*[host]*/,
                                        const float *radius_avg /*[host]*/, int nclass, int *kept_idxs, int *kept_offsets,
                                        float *kept_centers, int *prim_idxs, int *prim_offsets, float *prim_centers,
                                        int *frag_idxs, int *frag_offsets, float *frag_centers, int *post_idxs,
                                        int *post_offsets, int *counts /*[host,8]*/, void *workspace,
                                        size_t workspace_bytes, ms3d_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    for (int i = 0; i < 8; i++) counts[i] = 0;
    MS3D_CHECK(hipMemsetAsync(kept_offsets, 0, sizeof(int), stream));
    MS3D_CHECK(hipMemsetAsync(prim_offsets, 0, sizeof(int), stream));
    MS3D_CHECK(hipMemsetAsync(frag_offsets, 0, sizeof(int), stream));
    MS3D_CHECK(hipMemsetAsync(post_offsets, 0, sizeof(int), stream));
    if (N <= 0) return 0;
    HaWs w;
    if (carve(w, N, nclass, workspace) > workspace_bytes) return MS3D_E_WORKSPACE;
    MS3D_CHECK(hipMemcpyAsync(w.avg, point_num_avg, sizeof(float) * nclass, hipMemcpyHostToDevice, stream));
    MS3D_CHECK(hipMemcpyAsync(w.avg + nclass, radius_avg, sizeof(float) * nclass, hipMemcpyHostToDevice, stream));
    int cc_counts[2];
    int rc = ms3d_bfs_run_internal(0, 0, 0.f, capped_hint, semantic_label, ball_query_idxs, n_edges, start_len, N, w.cc_idx,
                                   w.cc_off, cc_counts, w.bfs_ws, w.bfs_bytes, stream);
    if (rc) return rc;
    const int ncc = cc_counts[0];
    const int nb = ms3d_divup(ncc, 128);
    ha_describe_kernel<<<ms3d_divup((long)ncc * 64, 256), 256, 0, stream>>>(ncc, w.cc_idx, w.cc_off, semantic_label,
                                                                          coord_shift, batch_idxs, w.avg, w.cc);
    MS3D_LAUNCH_CHECK();
    if (using_set_aggr) {
        ha_nearest_kernel<<<nb, 128, 0, stream>>>(ncc, w.cc, w.avg + nclass, w.absorb_to);
        MS3D_LAUNCH_CHECK();
        MS3D_CHECK(hipMemsetAsync(post_idxs, 0, sizeof(int) * 2 * (size_t)N, stream));
    }
    ha_sizes_kernel<<<nb, 128, 0, stream>>>(ncc, using_set_aggr, w.cc, w.absorb_to, w.kept_flag, w.kept_rows, w.prim_flag,
                                           w.prim_rows);
    MS3D_LAUNCH_CHECK();
    ha_frag_sizes_kernel<<<nb, 128, 0, stream>>>(ncc, w.cc, w.frag_flag, w.frag_rows, w.praw_rows);
    MS3D_LAUNCH_CHECK();
    if ((rc = ms3d_exclusive_scan_i32(w.kept_flag, w.kept_id, ncc, w.totals + 0, w.scan_ws, stream))) return rc;
    if ((rc = ms3d_exclusive_scan_i32(w.kept_rows, w.kept_off, ncc, w.totals + 1, w.scan_ws, stream))) return rc;
    if ((rc = ms3d_exclusive_scan_i32(w.prim_flag, w.prim_id, ncc, w.totals + 2, w.scan_ws, stream))) return rc;
    if ((rc = ms3d_exclusive_scan_i32(w.prim_rows, w.prim_off, ncc, w.totals + 3, w.scan_ws, stream))) return rc;
    if ((rc = ms3d_exclusive_scan_i32(w.frag_flag, w.frag_id, ncc, w.totals + 4, w.scan_ws, stream))) return rc;
    if ((rc = ms3d_exclusive_scan_i32(w.frag_rows, w.frag_off, ncc, w.totals + 5, w.scan_ws, stream))) return rc;
    if ((rc = ms3d_exclusive_scan_i32(w.praw_rows, w.praw_off, ncc, w.totals + 6, w.scan_ws, stream))) return rc;
    ha_parts_kernel<<<ncc, 128, 0, stream>>>(ncc, using_set_aggr, w.cc, w.cc_idx, w.cc_off, w.absorb_to, w.kept_id,
                                            w.kept_off, w.prim_id, w.prim_off, w.frag_id, w.frag_off, w.praw_off, kept_idxs,
                                            kept_offsets, kept_centers, prim_idxs, prim_offsets, prim_centers, frag_idxs,
                                            frag_offsets, frag_centers, post_idxs, post_offsets);
    MS3D_LAUNCH_CHECK();
    int h[8] = {0};
    MS3D_CHECK(hipMemcpyAsync(h, w.totals, sizeof(int) * 7, hipMemcpyDeviceToHost, stream));
    MS3D_CHECK(hipStreamSynchronize(stream));
    for (int i = 0; i < 7; i++) counts[i] = h[i];
    return 0;
}

}  // extern "C"
