/*
 * sparse_oracle.c -- TEST INFRASTRUCTURE ONLY (never imported by the product path).
 *
 * Plain-C CPU restatement of the MinkowskiEngine subset the reference calls
 * (NVIDIA/MinkowskiEngine, un-pinned: reference README.md:45 installs master, README.md:73
 * the PyPI 0.5.4 release; the package is NOT under /root/reference and not installable
 * here).  PARITY UNPINNED against ME itself: there is no ME build, golden vector or test
 * in the reference for this boundary.  The restatement follows ME v0.5.4's published
 * semantics as used at the reference call sites
 *   minsu3d/model/module/backbone.py:14-17,38   common.py:27-40,67-69,75-77,93
 *   minsu3d/model/module/tiny_unet.py:13-15     general_model.py:187-191
 *   minsu3d/data/data_module.py:94-96           data/dataset/general_dataset.py:159-163
 * and is pinned instead against dense torch.nn.functional.conv3d / conv_transpose3d on
 * densified inputs (tests/test_sparse_cpu.py).
 *
 * Semantics restated:
 *   sparse_quantize : unique rows of int coords, first occurrence wins, first-occurrence order
 *   k3 s1 conv      : out[i] = sum_k in[j(i,k)] @ W[k],  c_j = c_i + o_k*ts,
 *                     o_k = (ix-1, iy-1, iz-1), k = ix + 3*iy + 9*iz   (x fastest)
 *   k2 s2 conv      : out coords = unique(floor(c / 2ts) * 2ts) in first-occurrence order,
 *                     k = ix + 2*iy + 4*iz with (ix,iy,iz) = (c - c_out)/ts in {0,1}^3
 *   k2 s2 transpose : out[fine] = in[parent(fine)] @ W[k(fine)] on the cached fine coord set
 * Every conv is expressed through one "neighbour table" nbr[V_out][K] (input row or -1),
 * the same representation the HIP engine consumes.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -fPIC -shared (oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <omp.h>
typedef uint64_t u64;

/* bound the OpenMP team (hosts with hundreds of cores thrash on these short loops) */
void orc_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }

static inline u64 pack_key(int b, int x, int y, int z)
{
    return ((u64)(uint32_t)(b & 0x7FFFF) << 45) | ((u64)(uint32_t)((x + 16384) & 0x7FFF) << 30) |
           ((u64)(uint32_t)((y + 16384) & 0x7FFF) << 15) | (u64)(uint32_t)((z + 16384) & 0x7FFF);
}
static inline int in_range(int x) { return x >= -16384 && x < 16384; }
static inline u64 mix(u64 k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return k;
}
typedef struct { u64 *keys; int *vals; u64 mask; } hmap;
#define EMPTY_KEY 0xFFFFFFFFFFFFFFFFULL
static void hmap_init(hmap *h, int n)
{
    u64 cap = 16;
    while (cap < (u64)n * 2 + 2) cap <<= 1;
    h->keys = (u64 *)malloc(sizeof(u64) * cap);
    h->vals = (int *)malloc(sizeof(int) * cap);
    memset(h->keys, 0xFF, sizeof(u64) * cap);
    h->mask = cap - 1;
}
static void hmap_free(hmap *h) { free(h->keys); free(h->vals); }
/* insert if absent; returns stored value (first writer wins) */
static int hmap_put(hmap *h, u64 key, int val)
{
    u64 s = mix(key) & h->mask;
    for (;;) {
        if (h->keys[s] == EMPTY_KEY) { h->keys[s] = key; h->vals[s] = val; return val; }
        if (h->keys[s] == key) return h->vals[s];
        s = (s + 1) & h->mask;
    }
}
static int hmap_get(const hmap *h, u64 key)
{
    u64 s = mix(key) & h->mask;
    for (;;) {
        if (h->keys[s] == EMPTY_KEY) return -1;
        if (h->keys[s] == key) return h->vals[s];
        s = (s + 1) & h->mask;
    }
}

/* ME.utils.sparse_quantize on integer rows [N,4] (b,x,y,z): first-occurrence unique.
 * unique_idx[u] = first row with that coordinate (ascending), inverse[i] = u.
 * Returns the number of unique rows. */
int orc_sparse_quantize(const int *coords, int n, int *unique_idx, int *inverse)
{
    hmap h; hmap_init(&h, n);
    int nu = 0;
    for (int i = 0; i < n; i++) {
        const int *c = coords + (size_t)i * 4;
        int got = hmap_put(&h, pack_key(c[0], c[1], c[2], c[3]), nu);
        if (got == nu) unique_idx[nu++] = i;
        inverse[i] = got;
    }
    hmap_free(&h);
    return nu;
}

/* submanifold 3x3x3 table: nbr[i*27+k] = row j with c_j = c_i + o_k*ts, else -1 */
void orc_kmap_k3(const int *coords, int V, int ts, int *nbr)
{
    hmap h; hmap_init(&h, V);
    for (int i = 0; i < V; i++) {
        const int *c = coords + (size_t)i * 4;
        hmap_put(&h, pack_key(c[0], c[1], c[2], c[3]), i);
    }
    for (int i = 0; i < V; i++) {
        const int *c = coords + (size_t)i * 4;
        for (int k = 0; k < 27; k++) {
            const int x = c[1] + (k % 3 - 1) * ts, y = c[2] + ((k / 3) % 3 - 1) * ts,
                      z = c[3] + (k / 9 - 1) * ts;
            nbr[(size_t)i * 27 + k] =
                (in_range(x) && in_range(y) && in_range(z)) ? hmap_get(&h, pack_key(c[0], x, y, z)) : -1;
        }
    }
    hmap_free(&h);
}

/* stride-2 downsample.  out_coords [Vc,4] = floor(c/(2ts))*2ts in first-occurrence order,
 * parent[i] = coarse row of fine row i, koff[i] = ix + 2*iy + 4*iz.  Returns Vc. */
int orc_downsample(const int *coords, int V, int ts, int *out_coords, int *parent, int *koff)
{
    hmap h; hmap_init(&h, V);
    const int t2 = ts * 2;
    int nc = 0;
    for (int i = 0; i < V; i++) {
        const int *c = coords + (size_t)i * 4;
        int q[3];
        for (int d = 0; d < 3; d++) {
            int v = c[d + 1];
            int fl = (v >= 0) ? (v / t2) : -((-v + t2 - 1) / t2);
            q[d] = fl * t2;
        }
        int got = hmap_put(&h, pack_key(c[0], q[0], q[1], q[2]), nc);
        if (got == nc) {
            out_coords[(size_t)nc * 4 + 0] = c[0];
            out_coords[(size_t)nc * 4 + 1] = q[0];
            out_coords[(size_t)nc * 4 + 2] = q[1];
            out_coords[(size_t)nc * 4 + 3] = q[2];
            nc++;
        }
        parent[i] = got;
        koff[i] = (c[1] - q[0]) / ts + 2 * ((c[2] - q[1]) / ts) + 4 * ((c[3] - q[2]) / ts);
    }
    hmap_free(&h);
    return nc;
}

/* k2s2 table  nbr_down[p*8+k] = fine row with parent p and offset k (or -1);
 * transposed table nbr_up[f*8+k] = (k == koff[f]) ? parent[f] : -1 */
void orc_kmap_k2(const int *parent, const int *koff, int Vf, int Vc, int *nbr_down, int *nbr_up)
{
    for (size_t t = 0; t < (size_t)Vc * 8; t++) nbr_down[t] = -1;
    for (size_t t = 0; t < (size_t)Vf * 8; t++) nbr_up[t] = -1;
    for (int f = 0; f < Vf; f++) {
        nbr_down[(size_t)parent[f] * 8 + koff[f]] = f;
        nbr_up[(size_t)f * 8 + koff[f]] = parent[f];
    }
}

/* out[i,:] = sum_k in[nbr[i,k],:] @ W[k]   (W [K,Cin,Cout]); k ascending, c ascending */
void orc_conv_fwd(const float *in, const float *W, const int *nbr, int Vout, int K, int Cin,
                  int Cout, float *out)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < Vout; i++) {
        float *o = out + (size_t)i * Cout;
        for (int j = 0; j < Cout; j++) o[j] = 0.f;
        for (int k = 0; k < K; k++) {
            const int r = nbr[(size_t)i * K + k];
            if (r < 0) continue;
            const float *x = in + (size_t)r * Cin;
            const float *w = W + (size_t)k * Cin * Cout;
            for (int c = 0; c < Cin; c++) {
                const float xv = x[c];
                for (int j = 0; j < Cout; j++) o[j] += xv * w[(size_t)c * Cout + j];
            }
        }
    }
}

/* din[r,:] += dout[i,:] @ W[k]^T for every (i,k) with nbr[i,k] = r.  din pre-zeroed. */
void orc_conv_bwd_data(const float *dout, const float *W, const int *nbr, int Vout, int K, int Cin,
                       int Cout, float *din)
{
    for (int i = 0; i < Vout; i++) {
        const float *g = dout + (size_t)i * Cout;
        for (int k = 0; k < K; k++) {
            const int r = nbr[(size_t)i * K + k];
            if (r < 0) continue;
            float *d = din + (size_t)r * Cin;
            const float *w = W + (size_t)k * Cin * Cout;
            for (int c = 0; c < Cin; c++) {
                float s = 0.f;
                for (int j = 0; j < Cout; j++) s += g[j] * w[(size_t)c * Cout + j];
                d[c] += s;
            }
        }
    }
}

/* dW[k] = sum_i in[nbr[i,k],:]^T dout[i,:]   (accumulated in double, stored f32) */
void orc_conv_bwd_weight(const float *in, const float *dout, const int *nbr, int Vout, int K,
                         int Cin, int Cout, float *dW)
{
#pragma omp parallel for schedule(static)
    for (int k = 0; k < K; k++) {
        double *acc = (double *)calloc((size_t)Cin * Cout, sizeof(double));
        for (int i = 0; i < Vout; i++) {
            const int r = nbr[(size_t)i * K + k];
            if (r < 0) continue;
            const float *x = in + (size_t)r * Cin;
            const float *g = dout + (size_t)i * Cout;
            for (int c = 0; c < Cin; c++)
                for (int j = 0; j < Cout; j++) acc[(size_t)c * Cout + j] += (double)x[c] * (double)g[j];
        }
        for (size_t t = 0; t < (size_t)Cin * Cout; t++) dW[(size_t)k * Cin * Cout + t] = (float)acc[t];
        free(acc);
    }
}
