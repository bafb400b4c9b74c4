/*
 * grouping_oracle.c -- TEST INFRASTRUCTURE ONLY (never imported by the product path).
 *
 * Plain-C CPU restatement of the reference's `minsu3d/common_ops` grouping path.
 * Each function cites the reference file:line (relative to /root/reference) it follows.
 * It is the parity checker for the HIP kernels in minsu3d_amd/csrc/ and the
 * `cpu_baseline` leg of bench.py.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline may load it.
 *
 * Pinning status: pg_bfs_cluster / sg_bfs_cluster / hierarchical_aggregation (CC+split)
 * are checked against the reference's own C++ (built into oracle/_ref, see
 * oracle/build_ref.py) and against the tests/golden fixtures generated from it.  The GPU-only
 * reference ops (ball query, segment ops, pools, IoU) are restated line by line from
 * the reference .cu files and cross-checked against the reference kernels themselves
 * when oracle/_ref runs on a GPU box (tests/test_grouping_gpu.py::test_reference_gpu_kernels_agree).
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (see oracle/Makefile).  Contraction is
 * OFF so that every fused multiply-add below is an explicit fmaf().
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BQ_CAP 1000 /* bfs_cluster.cu:21,38 -- per-point neighbour cap */

/* ---------------------------------------------------------------- ball query
 * minsu3d/common_ops/src/bfs_cluster/bfs_cluster.cu:15-60 (kernel) and the retry loop
 * of functions/common_ops.py:31-40.  Canonical form: start = exclusive
 * prefix sum of len in point order (the reference's atomicAdd start is an arbitrary
 * permutation), lists ascending, self included, strict d2 < r*r, len = min(hits,1000).
 * d2 is pinned to fmaf(dz,dz,fmaf(dy,dy,dx*dx)) (device compilers contract the
 * reference expression at bfs_cluster.cu:36 to exactly this chain).
 *
 * Pass idx == NULL to only count.  Returns nActive = sum(len).  `cap` is the number of
 * ints available in idx; entries past it are dropped like bfs_cluster.cu:51-58.      */
static int bq_one(int i, float r2, const float *xyz, const uint8_t *batch_idxs, const int *batch_offsets,
                  int *out /* may be NULL */, long room)
{
    const float ox = xyz[i * 3 + 0], oy = xyz[i * 3 + 1], oz = xyz[i * 3 + 2];
    const int b = batch_idxs[i];
    const int s = batch_offsets[b], e = batch_offsets[b + 1]; /* :28-30 */
    int cnt = 0;
    for (int k = s; k < e; k++) { /* :32-46 */
        const float dx = ox - xyz[k * 3 + 0];
        const float dy = oy - xyz[k * 3 + 1];
        const float dz = oz - xyz[k * 3 + 2];
        const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
        if (d2 < r2) {
            if (cnt < BQ_CAP) {
                if (out && cnt < room) out[cnt] = k;
            } else
                break;
            ++cnt;
        }
    }
    return cnt;
}

long orc_ballquery_batch_p(int n, float radius, const float *xyz, const uint8_t *batch_idxs,
                           const int *batch_offsets, int *idx, long cap, int *start_len)
{
    const float r2 = radius * radius; /* bfs_cluster.cu:23 */
    /* points are independent: the per-point scans run on all host cores, the canonical starts are the
     * serial prefix sum of the counts */
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < n; i++) start_len[i * 2 + 1] = bq_one(i, r2, xyz, batch_idxs, batch_offsets, NULL, 0);
    long cum = 0;
    for (int i = 0; i < n; i++) {
        start_len[i * 2 + 0] = (int)cum;
        cum += start_len[i * 2 + 1];
    }
    if (idx) {
#pragma omp parallel for schedule(dynamic, 64)
        for (int i = 0; i < n; i++) {
            const long st = start_len[i * 2];
            if (st < cap) bq_one(i, r2, xyz, batch_idxs, batch_offsets, idx + st, cap - st);
        }
    }
    return cum;
}

/* ---------------------------------------------------------------- BFS clustering
 * bfs_cluster.cpp:28-54 (pg_find_cc), :56-80 (sg_find_cc), :86-101 / :103-129
 * (get_clusters), :131-140 (fill_cluster_idxs_).
 * mode 0 = pg (label test + int threshold), mode 1 = sg (no label test, float thr).
 * Outputs: cluster_idxs [sum,2] (cluster_id, point), cluster_offsets [nCluster+1].
 * Caller provides capacity n rows / n+1 offsets.  Returns nCluster; *sum_out = rows.  */
static int bfs_generic(int mode, const int16_t *sem, const int *ball_idx, const int *start_len,
                       int n, float thr_f, int thr_i, int *cluster_idxs, int *cluster_offsets,
                       int *sum_out, const float *coord_shift, const uint8_t *batch_idxs,
                       float *acc_xyz /* per cluster 3 */, int16_t *cl_label, int16_t *cl_batch,
                       int keep_all)
{
    int *visited = (int *)calloc((size_t)(n > 0 ? n : 1), sizeof(int));
    int *queue = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    int ncl = 0, sum = 0;
    cluster_offsets[0] = 0;
    for (int i = 0; i < n; i++) {
        if (visited[i]) continue;
        int head = 0, tail = 0;
        float ax = 0.f, ay = 0.f, az = 0.f;
        queue[tail++] = i;
        visited[i] = 1;
        if (coord_shift) { /* hierarchical_aggregation.cpp:13-16 */
            ax += coord_shift[i * 3 + 0];
            ay += coord_shift[i * 3 + 1];
            az += coord_shift[i * 3 + 2];
        }
        while (head < tail) {
            const int cur = queue[head++];
            const int s = start_len[cur * 2], len = start_len[cur * 2 + 1];
            const int16_t lc = sem ? sem[cur] : 0;
            for (int t = s; t < s + len; t++) {
                const int j = ball_idx[t];
                if (mode == 0 && sem[j] != lc) continue; /* bfs_cluster.cpp:44 */
                if (visited[j]) continue;
                visited[j] = 1;
                queue[tail++] = j;
                if (coord_shift) { /* hierarchical_aggregation.cpp:32-35 */
                    ax += coord_shift[j * 3 + 0];
                    ay += coord_shift[j * 3 + 1];
                    az += coord_shift[j * 3 + 2];
                }
            }
        }
        int keep;
        if (keep_all)
            keep = 1;
        else if (mode == 0)
            keep = tail >= thr_i; /* bfs_cluster.cpp:94 */
        else
            keep = (float)tail >= thr_f; /* bfs_cluster.cpp:121  (int)size >= thr, thr float */
        if (keep) {
            for (int q = 0; q < tail; q++) {
                cluster_idxs[(sum + q) * 2 + 0] = ncl;
                cluster_idxs[(sum + q) * 2 + 1] = queue[q];
            }
            if (acc_xyz) {
                acc_xyz[ncl * 3 + 0] = ax;
                acc_xyz[ncl * 3 + 1] = ay;
                acc_xyz[ncl * 3 + 2] = az;
            }
            if (cl_label) cl_label[ncl] = sem[i];
            if (cl_batch) cl_batch[ncl] = batch_idxs ? (int16_t)batch_idxs[i] : (int16_t)-1;
            sum += tail;
            ncl++;
            cluster_offsets[ncl] = sum;
        }
    }
    free(visited);
    free(queue);
    *sum_out = sum;
    return ncl;
}

int orc_pg_bfs_cluster(const int16_t *sem, const int *ball_idx, const int *start_len, int n,
                       int threshold, int *cluster_idxs, int *cluster_offsets, int *sum_out)
{
    return bfs_generic(0, sem, ball_idx, start_len, n, 0.f, threshold, cluster_idxs,
                       cluster_offsets, sum_out, NULL, NULL, NULL, NULL, NULL, 0);
}

int orc_sg_bfs_cluster(const float *class_numpoint_mean, const int *ball_idx,
                       const int *start_len, int n, float threshold, int class_id,
                       int *cluster_idxs, int *cluster_offsets, int *sum_out)
{
    /* bfs_cluster.cpp:113-120 */
    const float m = class_numpoint_mean[class_id];
    const float thr = (m == -1.f) ? threshold : threshold * m;
    return bfs_generic(1, NULL, ball_idx, start_len, n, thr, 0, cluster_idxs, cluster_offsets,
                       sum_out, NULL, NULL, NULL, NULL, NULL, 0);
}

/* ---------------------------------------------------------------- segment ops
 * sec_mean.cu:12-27, :38-53, :64-79.  Sequential in row order, divide-then-add.     */
void orc_sec_mean(int P, int C, const float *inp, const int *offsets, float *out)
{
    for (int p = 0; p < P; p++) {
        const int s = offsets[p], e = offsets[p + 1];
        const float count = (float)(e - s);
        for (int c = 0; c < C; c++) {
            float mean = 0.f;
            for (int i = s; i < e; i++) mean += inp[(size_t)i * C + c] / count;
            out[(size_t)p * C + c] = mean;
        }
    }
}
void orc_sec_min(int P, int C, const float *inp, const int *offsets, float *out)
{
    for (int p = 0; p < P; p++)
        for (int c = 0; c < C; c++) {
            float v = INFINITY; /* 1e50 -> +inf in f32, sec_mean.cu:44 */
            for (int i = offsets[p]; i < offsets[p + 1]; i++)
                if (inp[(size_t)i * C + c] < v) v = inp[(size_t)i * C + c];
            out[(size_t)p * C + c] = v;
        }
}
void orc_sec_max(int P, int C, const float *inp, const int *offsets, float *out)
{
    for (int p = 0; p < P; p++)
        for (int c = 0; c < C; c++) {
            float v = -INFINITY;
            for (int i = offsets[p]; i < offsets[p + 1]; i++)
                if (inp[(size_t)i * C + c] > v) v = inp[(size_t)i * C + c];
            out[(size_t)p * C + c] = v;
        }
}

/* ---------------------------------------------------------------- pools
 * roipool.cu:12-31 (fp), :42-49 (bp), :60-80 (avg fp), :94-108 (avg bp).
 * Backward accumulates in proposal order (the reference's atomicAdd order is
 * arbitrary; each d_feats element receives exactly one addend for avg-pool and for
 * roipool whenever proposals are disjoint, so results are order independent there). */
void orc_roipool_fp(int P, int C, const float *feats, const int *offsets, float *out, int *maxidx)
{
    for (int p = 0; p < P; p++)
        for (int c = 0; c < C; c++) {
            int am = -1;
            float v = -INFINITY;
            for (int i = offsets[p]; i < offsets[p + 1]; i++)
                if (feats[(size_t)i * C + c] > v) {
                    am = i;
                    v = feats[(size_t)i * C + c];
                }
            maxidx[(size_t)p * C + c] = am;
            out[(size_t)p * C + c] = v;
        }
}
void orc_roipool_bp(int P, int C, float *d_feats, const int *offsets, const int *maxidx,
                    const float *d_out)
{
    (void)offsets;
    for (int p = 0; p < P; p++)
        for (int c = 0; c < C; c++) {
            const int am = maxidx[(size_t)p * C + c];
            d_feats[(size_t)am * C + c] += d_out[(size_t)p * C + c];
        }
}
void orc_global_avg_pool_fp(int P, int C, const float *feats, const int *offsets, float *out)
{
    for (int p = 0; p < P; p++) {
        const int s = offsets[p], e = offsets[p + 1];
        for (int c = 0; c < C; c++) {
            float v = 0.f;
            for (int i = s; i < e; i++) v += feats[(size_t)i * C + c];
            out[(size_t)p * C + c] = v / (float)(e - s);
        }
    }
}
void orc_global_avg_pool_bp(int P, int C, float *d_feats, const int *offsets, const float *d_out)
{
    for (int p = 0; p < P; p++) {
        const int s = offsets[p], e = offsets[p + 1];
        for (int c = 0; c < C; c++)
            for (int i = s; i < e; i++)
                d_feats[(size_t)i * C + c] += d_out[(size_t)p * C + c] / (float)(e - s);
    }
}

/* ---------------------------------------------------------------- IoU family
 * get_iou.cu:12-29 == cal_iou_and_masklabel.cu:14-38 (on_cluster); :40-71 (on_pred);
 * :73-105 (mask label).  The 1e-5 literal is a double: divide in double, store f32. */
static void iou_generic(int I, int P, const int *prop_idx, const int *prop_off,
                        const int16_t *inst_labels, const int *inst_pointnum, float *iou,
                        const float *sigmoid)
{
    int *cnt = (int *)malloc(sizeof(int) * (size_t)(I > 0 ? I : 1));
    for (int p = 0; p < P; p++) {
        const int s = prop_off[p], e = prop_off[p + 1];
        int total = 0;
        memset(cnt, 0, sizeof(int) * (size_t)I);
        for (int i = s; i < e; i++) {
            if (sigmoid && !(sigmoid[i] > 0.5)) continue; /* .cu:51,63 */
            total++;
            const int lab = inst_labels[prop_idx[i]];
            if (lab >= 0 && lab < I) cnt[lab]++;
        }
        for (int k = 0; k < I; k++) {
            const int inter = cnt[k];
            iou[(size_t)p * I + k] =
                (float)((float)inter / ((float)(total + inst_pointnum[k] - inter) + 1e-5));
        }
    }
    free(cnt);
}
void orc_get_iou(int I, int P, const int *prop_idx, const int *prop_off,
                 const int16_t *inst_labels, const int *inst_pointnum, float *iou)
{
    iou_generic(I, P, prop_idx, prop_off, inst_labels, inst_pointnum, iou, NULL);
}
void orc_get_mask_iou_on_cluster(int I, int P, const int *prop_idx, const int *prop_off,
                                 const int16_t *inst_labels, const int *inst_pointnum, float *iou)
{
    iou_generic(I, P, prop_idx, prop_off, inst_labels, inst_pointnum, iou, NULL);
}
void orc_get_mask_iou_on_pred(int I, int P, const int *prop_idx, const int *prop_off,
                              const int16_t *inst_labels, const int *inst_pointnum, float *iou,
                              const float *sigmoid)
{
    iou_generic(I, P, prop_idx, prop_off, inst_labels, inst_pointnum, iou, sigmoid);
}
void orc_get_mask_label(int I, int P, int ignored_label, float iou_thr, const int *prop_idx,
                        const int *prop_off, const int16_t *inst_labels, const int16_t *inst_cls,
                        const float *iou, uint8_t *mask_label, uint8_t *mask_label_mask)
{
    for (int p = 0; p < P; p++) {
        float max_iou = 0.f;
        int max_ind = 0;
        for (int k = 0; k < I; k++) /* .cu:84-92 */
            if (iou[(size_t)p * I + k] > max_iou && inst_cls[k] != ignored_label) {
                max_iou = iou[(size_t)p * I + k];
                max_ind = k;
            }
        if (max_iou >= iou_thr) /* .cu:95-103 */
            for (int i = prop_off[p]; i < prop_off[p + 1]; i++) {
                if (inst_labels[prop_idx[i]] == max_ind) mask_label[i] = 1;
                mask_label_mask[i] = 1;
            }
    }
}

/* ---------------------------------------------------------------- HAIS
 * hierarchical_aggregation.cpp:8-40 (find_cc), :43-78 (split_clusters), :81-97
 * (fill_cluster_idxs_), .cu:20-64 (fragment_find_primary_), :69-91 (concat_fragments_),
 * :158-179 (host merge), plus functions/hais_ops.py:55-73 (kept-then-primary concat).
 * Absorbed fragments are appended in ascending fragment index (canonical; the
 * reference order is atomic-dependent).
 * Outputs have capacity n rows / n+1 offsets.  Returns nCluster.                     */
#define HA_MAX_FRAG 1024
#define HA_MAX_PTS 8192
/* the four lists + centre rows (x, y, z, class, scene) hierarchical_aggregation.cpp:133-175 hands back: all fragments
 * (size < 0.3 avg), kept fragments (>= 0.05 avg), primaries, primaries with their absorbed fragments (set aggregation) */
typedef struct {
    int *frag_idx, *frag_off; float *frag_ctr; int n_frag, frag_sum;
    int *kept_idx, *kept_off; float *kept_ctr; int n_kept, kept_sum;
    int *prim_idx, *prim_off; float *prim_ctr; int n_prim, prim_sum;
    int *post_idx, *post_off; int n_post, post_sum;
} orc_ha_parts;

/* append the components with `kind & bit` (ascending seed) to a (cluster id, point) list; with `absorb_to`, a primary
 * is followed by the fragments absorbed into it in ascending fragment index under the caps of .cu:80-90 */
static void ha_emit(int ncc, const char *kind, int bit, const int *all_idx, const int *all_off, const int *absorb_to,
                    int unused, int *out_idx, int *out_off, int *ncl_io, int *sum_io, float *out_ctr, const float *ctr)
{
    (void)unused;
    int ncl = *ncl_io, sum = *sum_io;
    for (int c = 0; c < ncc; c++) {
        if (!(kind[c] & bit)) continue;
        for (int q = all_off[c]; q < all_off[c + 1]; q++) {
            out_idx[sum * 2 + 0] = ncl;
            out_idx[sum * 2 + 1] = all_idx[q * 2 + 1];
            sum++;
        }
        if (absorb_to) {
            int nfrag = 0, npts = 0;
            for (int f = 0; f < ncc; f++) {
                if (absorb_to[f] != c) continue;
                if (nfrag >= HA_MAX_FRAG) break;
                nfrag++;
                for (int q = all_off[f]; q < all_off[f + 1]; q++) {
                    if (npts < HA_MAX_PTS) {
                        out_idx[sum * 2 + 0] = ncl;
                        out_idx[sum * 2 + 1] = all_idx[q * 2 + 1];
                        sum++;
                        npts++;
                    }
                }
            }
        }
        if (out_ctr)
            for (int k = 0; k < 5; k++) out_ctr[ncl * 5 + k] = ctr[c * 5 + k];
        out_off[++ncl] = sum;
    }
    *ncl_io = ncl;
    *sum_io = sum;
}

static int ha_run(const int16_t *sem, const float *coord_shift,
                  const uint8_t *batch_idxs, const int *ball_idx,
                  const int *start_len, int n, int using_set_aggr,
                  const float *point_num_avg, const float *radius_avg,
                  int *cluster_idxs, int *cluster_offsets, int *sum_out, orc_ha_parts *parts)
{
    int *all_idx = (int *)malloc(sizeof(int) * 2 * (size_t)(n > 0 ? n : 1));
    int *all_off = (int *)malloc(sizeof(int) * (size_t)(n + 1));
    float *acc = (float *)malloc(sizeof(float) * 3 * (size_t)(n > 0 ? n : 1));
    int16_t *lab = (int16_t *)malloc(sizeof(int16_t) * (size_t)(n > 0 ? n : 1));
    int16_t *bat = (int16_t *)malloc(sizeof(int16_t) * (size_t)(n > 0 ? n : 1));
    int sum_all = 0;
    const int ncc = bfs_generic(0, sem, ball_idx, start_len, n, 0.f, 0, all_idx, all_off, &sum_all,
                                coord_shift, batch_idxs, acc, lab, bat, 1);
    /* classify, split_clusters :55-75 */
    char *kind = (char *)malloc((size_t)(ncc > 0 ? ncc : 1)); /* bit0 fragment, bit1 kept, bit2 primary */
    float *ctr = (float *)malloc(sizeof(float) * 5 * (size_t)(ncc > 0 ? ncc : 1));
    for (int c = 0; c < ncc; c++) {
        const int size = all_off[c + 1] - all_off[c];
        const float mean = point_num_avg[lab[c]];
        const float low = (float)(0.05 * (double)mean);
        const float high = (float)(0.3 * (double)mean);
        kind[c] = 0;
        if ((float)size < high) {
            kind[c] |= 1;
            if ((float)size >= low) kind[c] |= 2;
        } else
            kind[c] |= 4;
        ctr[c * 5 + 0] = acc[c * 3 + 0] / (float)size; /* :84-88 */
        ctr[c * 5 + 1] = acc[c * 3 + 1] / (float)size;
        ctr[c * 5 + 2] = acc[c * 3 + 2] / (float)size;
        ctr[c * 5 + 3] = (float)lab[c];
        ctr[c * 5 + 4] = (float)bat[c];
    }
    /* fragment -> nearest primary, .cu:27-63 */
    int *absorb_to = (int *)malloc(sizeof(int) * (size_t)(ncc > 0 ? ncc : 1));
    for (int c = 0; c < ncc; c++) absorb_to[c] = -1;
    if (using_set_aggr) {
        for (int f = 0; f < ncc; f++) {
            if (!(kind[f] & 1)) continue;
            float nearest = 10000.f;
            int ni = -1;
            for (int p = 0; p < ncc; p++) {
                if (!(kind[p] & 4)) continue;
                if (fabsf(ctr[p * 5 + 3] - ctr[f * 5 + 3]) > 0.1) continue;
                if (fabsf(ctr[p * 5 + 4] - ctr[f * 5 + 4]) > 0.1) continue;
                const float dx = ctr[p * 5 + 0] - ctr[f * 5 + 0];
                const float dy = ctr[p * 5 + 1] - ctr[f * 5 + 1];
                const float dz = ctr[p * 5 + 2] - ctr[f * 5 + 2];
                const float d = dx * dx + dy * dy + dz * dz;
                if (d < nearest) {
                    nearest = d;
                    ni = p;
                }
            }
            if (ni < 0) continue;
            const int pn = all_off[ni + 1] - all_off[ni];
            const float r_size = (float)(0.01 * (double)sqrtf((float)pn));
            const float r_cls = radius_avg[(int)ctr[f * 5 + 3]];
            const float r_set = r_size > r_cls ? r_size : r_cls;
            if (nearest < r_set * r_set) absorb_to[f] = ni;
        }
    }
    /* emit: kept fragments first, then primaries (+absorbed), hais_ops.py:55-73 */
    int ncl = 0, sum = 0;
    cluster_offsets[0] = 0;
    ha_emit(ncc, kind, 2, all_idx, all_off, NULL, 0, cluster_idxs, cluster_offsets, &ncl, &sum, NULL, ctr);
    ha_emit(ncc, kind, 4, all_idx, all_off, using_set_aggr ? absorb_to : NULL, 0, cluster_idxs, cluster_offsets,
            &ncl, &sum, NULL, ctr);
    if (parts) { /* the lists hierarchical_aggregation.cpp:133-175 leaves in the caller's tensors, ids from 0 each */
        int c0 = 0, s0 = 0;
        parts->frag_off[0] = 0;
        ha_emit(ncc, kind, 1, all_idx, all_off, NULL, 0, parts->frag_idx, parts->frag_off, &c0, &s0, parts->frag_ctr, ctr);
        parts->n_frag = c0; parts->frag_sum = s0;
        c0 = s0 = 0; parts->kept_off[0] = 0;
        ha_emit(ncc, kind, 2, all_idx, all_off, NULL, 0, parts->kept_idx, parts->kept_off, &c0, &s0, parts->kept_ctr, ctr);
        parts->n_kept = c0; parts->kept_sum = s0;
        c0 = s0 = 0; parts->prim_off[0] = 0;
        ha_emit(ncc, kind, 4, all_idx, all_off, NULL, 0, parts->prim_idx, parts->prim_off, &c0, &s0, parts->prim_ctr, ctr);
        parts->n_prim = c0; parts->prim_sum = s0;
        c0 = s0 = 0; parts->post_off[0] = 0;
        if (using_set_aggr)
            ha_emit(ncc, kind, 4, all_idx, all_off, absorb_to, 0, parts->post_idx, parts->post_off, &c0, &s0, NULL, ctr);
        parts->n_post = c0; parts->post_sum = s0;
    }
    free(all_idx); free(all_off); free(acc); free(lab); free(bat); free(kind); free(ctr);
    free(absorb_to);
    *sum_out = sum;
    return ncl;
}

int orc_hierarchical_aggregation(const int16_t *sem, const float *coord_shift,
                                 const uint8_t *batch_idxs, const int *ball_idx,
                                 const int *start_len, int n, int using_set_aggr,
                                 const float *point_num_avg, const float *radius_avg,
                                 int *cluster_idxs, int *cluster_offsets, int *sum_out)
{
    return ha_run(sem, coord_shift, batch_idxs, ball_idx, start_len, n, using_set_aggr, point_num_avg, radius_avg,
                  cluster_idxs, cluster_offsets, sum_out, NULL);
}

/* as above, and the raw lists in `parts` (buffers: *_idx 2*2n ints, *_off n+1 ints, *_ctr 5n floats; post_idx 2*2n) */
int orc_hierarchical_aggregation_parts(const int16_t *sem, const float *coord_shift,
                                       const uint8_t *batch_idxs, const int *ball_idx,
                                       const int *start_len, int n, int using_set_aggr,
                                       const float *point_num_avg, const float *radius_avg,
                                       int *cluster_idxs, int *cluster_offsets, int *sum_out, orc_ha_parts *parts)
{
    return ha_run(sem, coord_shift, batch_idxs, ball_idx, start_len, n, using_set_aggr, point_num_avg, radius_avg,
                  cluster_idxs, cluster_offsets, sum_out, parts);
}
