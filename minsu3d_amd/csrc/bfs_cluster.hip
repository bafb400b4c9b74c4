// Order-exact breadth-first clustering on the GPU.  Replaces the reference's single-threaded host
// BFS (bfs_cluster/bfs_cluster.cpp:28-187: pg_find_cc / sg_find_cc / *_get_clusters /
// fill_cluster_idxs_) and the two PCIe round trips around it (model/pointgroup.py:49-55,62-65).
//
// The output must equal the serial FIFO BFS exactly: clusters ordered by seed =
// smallest not-yet-visited index, members in queue order, and -- when a neighbour list was cut at
// 1000 entries -- out-edge reachability on a DIRECTED graph.  Plan:
//
//   1. weak components  : union-find over all label-compatible out-edges, larger root hooked under
//                         smaller (root = smallest member): pre-hook under the smallest neighbour,
//                         a racy sampled link pass with plain stores, and a verify pass over all
//                         edges that takes an atomic only where two ends still disagree.  Clusters
//                         never cross a weak component, and a component smaller than the threshold
//                         cannot contain a surviving cluster, so only big components are expanded.
//   2. expansion        : three regimes, all reproducing the serial queue order (seeds in ascending
//                         index; members in (level, parent position, slot) order -- the order in
//                         which the serial loop would have pushed them, lists being ascending in j):
//      sparse graphs      persistent 512-thread workgroups pull components from a work list and
//                         replay the serial algorithm inside each (bfs_expand_kernel): per frontier
//                         chunk every out-edge is visited twice -- phase A posts atomicMin(claim[j], p),
//                         phase B lets the edge with claim[j] == p win, winners compacted by
//                         ballot / prefix sum;
//      dense symmetric    (no list at the 1000 cap) chip-wide level-synchronous mark / pull / win
//                         kernels over ALL components at once (glob_*);
//      dense directed     (capped lists) two chip-wide stages: the weak components expanded from their
//                         roots with pushed claims (dir_*), then the leftovers labelled by their
//                         smallest ancestor and expanded per label (dir2_*) -- see the comments there.
//   3. assembly         : per-seed sizes -> keep flags -> two exclusive scans give cluster ids
//                         (ascending seed) and output offsets; one pass copies members.
//
// All state lives in a caller-provided workspace; the only host traffic is the final 8-byte
// (nCluster, sumNPoint) read the caller needs to size its tensors.
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <atomic>
#include "common.h"
#include "scan.h"
#include "../../include/minsu3d_hip.h"

namespace {

constexpr int INT_BIG = 0x7fffffff;

struct Thr {
    int mode;     // 0 = pg (label test, int threshold), 1 = sg (no label test, float threshold),
                  // 2 = sg batched over classes: float threshold looked up through the seed's group id
    int thr_i;
    float thr_f;
    const uint8_t *group;     // mode 2: group (class*B + scene) of every point
    const float *thr_group;   // mode 2: threshold of every group
};
// `node` = the cluster's seed (any member works: a cluster never leaves its group)
__device__ __forceinline__ bool qualifies(const Thr &t, int size, int node)
{
    if (t.mode == 0) return size >= t.thr_i;                      // bfs_cluster.cpp:94
    if (t.mode == 1) return (float)size >= t.thr_f;                // bfs_cluster.cpp:121
    return (float)size >= t.thr_group[t.group[node]];
}

__device__ __forceinline__ int ld_agent(const int *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ int uf_find(int *parent, int x)
{
    int p = ld_agent(&parent[x]);
    while (p != x) {
        const int gp = ld_agent(&parent[p]);
        if (gp != p) __hip_atomic_store(&parent[x], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // path halving
        x = p;
        p = gp;
    }
    return x;
}
__device__ __forceinline__ void uf_union(int *parent, int a, int b)
{
    for (;;) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        const int hi = max(a, b), lo = min(a, b);
        if (atomicCAS(&parent[hi], hi, lo) == hi) return;
    }
}

// parent[] starts as a forest already: every point hangs under its smallest label-compatible neighbour (the first
// entry of its ascending list) when that index is below its own -- one coalesced read per point instead of the CAS
// storm of 232 k singletons meeting each other in the hook kernel
__global__ void bfs_init_kernel(int N, Thr thr, const int16_t *__restrict__ sem, const int *__restrict__ ball_idx,
                                const int *__restrict__ start_len, int *parent, int *comp_size, int *visited, int *claim,
                                int *cl_size, int *scratch_seed, int *defi, int *counters)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) {
        int p = i;
        if (start_len[i * 2 + 1] > 0) {
            const int j = ball_idx[start_len[i * 2]];
            if (j < i && (thr.mode != 0 || sem[j] == sem[i])) p = j;
        }
        parent[i] = p;
        comp_size[i] = 0;
        visited[i] = 0;
        claim[i] = INT_BIG;
        cl_size[i] = 0;
        scratch_seed[i] = -1;  // "slot not written": the assembly may run before an incomplete expansion is detected
        scratch_seed[N + i] = -1;
        defi[i] = 0;
    }
    if (i < 32) counters[i] = 0;
}

// Weak components in three steps (the atomics of a union cost ~30x a cached gather on this part -- 5.7 G/s against
// 240 G/s, tools/gather_micro.hip -- so the edges are looked at with plain loads and unions are kept for the few that
// need one):
//   link     : every point unites with two pseudo-randomly picked label-compatible neighbours.  In a dense
//              neighbourhood graph that alone connects almost every component (a random geometric graph of degree ~4).
//   compress : parent[i] = root(i) for every point.
//   verify   : one wave per point, lanes stride its list: an edge whose two ends carry the same (compressed) parent is
//              dismissed with one cached 4-byte gather; only the rest take the find / union path.  When the caller
//              knows the graph is symmetric (no list was cut at the cap) only the j < i half of every list is looked at:
//              the other half is the same undirected edge seen from the other end.
__global__ void bfs_link_kernel(int N, Thr thr, const int16_t *__restrict__ sem, const int *__restrict__ ball_idx,
                                const int *__restrict__ start_len, int *parent)
{
    // parent[] is compressed (every entry a root) when this kernel starts.  Plain, racy stores on purpose: a root is
    // only ever pointed at a SMALLER index (no cycle can form), a lost race loses one optional link, and the verify
    // pass (atomic unions) is what guarantees the result -- this pass only has to make its slow path rare.
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int st = start_len[i * 2], ln = start_len[i * 2 + 1];
    if (ln < 1) return;   // (a one-entry list is the point itself in a ball query's graph, but not in anybody's: linked like the rest)
    const unsigned h1 = ((unsigned)i * 0x9E3779B1u) >> 8, h2 = (((unsigned)i ^ 0x5bd1e995u) * 0x85ebca6bu) >> 8;
    const int j1 = ball_idx[st + (int)(h1 % (unsigned)ln)], j2 = ball_idx[st + (int)(h2 % (unsigned)ln)];
    const int lab = thr.mode == 0 ? (int)sem[i] : 0;
    const int ri = parent[i];
    const int r1 = (thr.mode != 0 || (int)sem[j1] == lab) ? parent[j1] : ri;
    const int r2 = (thr.mode != 0 || (int)sem[j2] == lab) ? parent[j2] : ri;
    const int lo = min(ri, min(r1, r2));
    if (ri != lo) parent[ri] = lo;
    if (r1 != lo) parent[r1] = lo;
    if (r2 != lo) parent[r2] = lo;
}

// parent[i] = root(i).  Plain loads and stores: a stale parent is still an ancestor (pointers only ever move towards
// the root) and roots do not change while this kernel runs.
__global__ void bfs_compress_kernel(int N, int *parent)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    int r = parent[i];
    for (;;) {
        const int g = parent[r];
        if (g == r) break;
        r = g;
    }
    parent[i] = r;
}

__global__ __launch_bounds__(256) void bfs_hook_kernel(int N, Thr thr, int symmetric, const int16_t *__restrict__ sem,
                                                       const int *__restrict__ ball_idx,
                                                       const int *__restrict__ start_len, int *parent)
{
    const int waves = blockDim.x >> 6;
    const int l = lane_id();
    // A point is a chain of dependent round trips (list header / own root -> list slices -> the neighbours' roots): the
    // header of the point after next and the first slices of the next point are requested while this point's gathers are
    // in flight (round 5: 250-315 us -> see profiles/r05_*; a stale own root only costs a find).
    const int stride = gridDim.x * waves;
    struct Hdr { int st, ln, lab, ri; };
    auto load_hdr = [&](int i) {
        Hdr h{0, 0, 0, 0};
        if (i < N) {
            h.st = start_len[i * 2]; h.ln = start_len[i * 2 + 1];
            h.lab = thr.mode == 0 ? (int)sem[i] : 0;
            h.ri = parent[i];
        }
        return h;
    };
    auto load_slices = [&](const Hdr &h, int (&j)[4]) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int t = 64 * u + l;
            j[u] = t < h.ln ? ball_idx[h.st + t] : INT_BIG;
        }
    };
    int i = blockIdx.x * waves + wave_id();
    Hdr hA = load_hdr(i), hB = load_hdr(i + stride);
    int jA[4];
    load_slices(hA, jA);
    for (; i < N; i += stride) {
        const Hdr hC = load_hdr(i + 2 * stride);
        int jB[4];
        load_slices(hB, jB);
        const int st = hA.st, ln = hA.ln, lab = hA.lab;
        int ri = hA.ri;         // compressed: the root as of the last compress pass
        // four 64-edge slices per trip: the index loads, then the parent gathers are in flight together
        // (a one-entry list of a graph a ball query vouches for is the point itself; of any other graph it may be an edge:
        // rounds 1-4 skipped it either way, which cut such points out of their component -- found by the random-digraph test)
        for (int t0 = 0; t0 < ln && ln > (symmetric ? 1 : 0); t0 += 256) {
            int j[4], pj[4];
            if (t0 == 0) {
#pragma unroll
                for (int u = 0; u < 4; u++) j[u] = jA[u];
            } else {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int t = t0 + 64 * u + l;
                    j[u] = t < ln ? ball_idx[st + t] : INT_BIG;
                }
            }
            // lists are ascending: once a slice starts at or above i the rest of a symmetric graph's list is the other
            // end's business
            if (symmetric && __shfl(j[0], 0, 64) >= i) break;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const bool look = j[u] != INT_BIG && j[u] != i && (!symmetric || j[u] < i);
                pj[u] = look ? parent[j[u]] : ri;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                bool merge = false;
                if (pj[u] != ri) merge = thr.mode != 0 || (int)sem[j[u]] == lab;  // bfs_cluster.cpp:44
                if (__ballot(merge) == 0ull) continue;
                // 64 lanes hooking onto the SAME root one CAS at a time serialise (one winner per round); instead every
                // lane resolves its own neighbour's root, the wave agrees on the smallest root in sight and each distinct
                // root is hooked under it: the CAS targets are distinct, one round in the common case
                const int rj = merge ? uf_find(parent, j[u]) : INT_BIG;
                const int rme = uf_find(parent, i);
                const int rmin = min(wave_min(rj), rme);
                if (merge && rj != rmin) uf_union(parent, rj, rmin);
                if (rme != rmin && l == 0) uf_union(parent, rme, rmin);
                const int rnew = uf_find(parent, i);  // refreshed for the next slices
#pragma unroll
                for (int v = 0; v < 4; v++)
                    if (v > u && pj[v] == ri) pj[v] = rnew;  // "already in my set" stays true under the new root
                ri = rnew;
            }
        }
        hA = hB; hB = hC;
#pragma unroll
        for (int u = 0; u < 4; u++) jA[u] = jB[u];
    }
}

// roots go to their own array: concurrent path halving may still rewrite parent[] entries with
// non-root ancestors while this kernel runs
__global__ void bfs_flatten_kernel(int N, int *parent, int *root, int *comp_size, const int *__restrict__ start_len,
                                   int *counters)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int r = uf_find(parent, i);
    root[i] = r;
    // wave-aggregated count: neighbouring points mostly share a root, one atomic per distinct root of the wave
    // (one atomic per point on a few hot counters took 230 us for 575k points)
    unsigned long long todo = __ballot(1);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int r0 = __shfl(r, leader, 64);
        const unsigned long long same = __ballot(r == r0) & todo;
        if (lane_id() == leader) atomicAdd(&comp_size[r0], __popcll(same));
        todo &= ~same;
    }
    // a list cut at the cap makes the graph directed (bfs_cluster.cu:38-43): only then can a weak component
    // hold more than one cluster, and only then is the order-by-replay kernel required
    if (start_len[i * 2 + 1] >= 1000 && counters[5] == 0) atomicOr(&counters[5], 1);
    // longer than the edge masks of the chip-wide directed expansion cover (never from a ball query: it stops at 1000)
    if (start_len[i * 2 + 1] > 1024 && counters[18] == 0) atomicOr(&counters[18], 1);
}

// strict mode, before anything dereferences the lists: every header inside the edge array, every target a point
__global__ void bfs_validate_kernel(int N, long n_edges, const int *__restrict__ ball_idx, const int *__restrict__ start_len,
                                    int *flag)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    bool bad = false;
    // both parts grid-stride: the grid is capped at 2^20 threads, N is not (headers past the cap went unchecked)
    for (long i = t; i < N; i += (long)gridDim.x * blockDim.x) {
        const long st = start_len[i * 2], ln = start_len[i * 2 + 1];
        bad |= st < 0 || ln < 0 || st + ln > n_edges;
    }
    for (long e = t; e < n_edges; e += (long)gridDim.x * blockDim.x) {
        const int j = ball_idx[e];
        bad |= j < 0 || j >= N;
    }
    if (__ballot(bad) && lane_id() == 0) *flag = 1;
}

__global__ void bfs_select_kernel(int N, Thr thr, const int *__restrict__ root, const int *__restrict__ comp_size,
                                  int *worklist, int *counters)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    if (root[i] == i && qualifies(thr, comp_size[i], i)) worklist[atomicAdd(&counters[0], 1)] = i;
}

template <int NTHREADS>
__device__ __forceinline__ int block_excl_scan_512(int v, int *total, int *s_wave)
{
    const int incl = wave_incl_scan(v);
    if (lane_id() == 63) s_wave[wave_id()] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NTHREADS / 64; w++) {
        const int t = s_wave[w];
        if (w < wave_id()) base += t;
        tot += t;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}
template <int NTHREADS>
__device__ __forceinline__ int block_min_512(int v, int *s_wave)
{
    v = wave_min(v);
    if (lane_id() == 0) s_wave[wave_id()] = v;
    __syncthreads();
    int m = INT_BIG;
#pragma unroll
    for (int w = 0; w < NTHREADS / 64; w++) m = min(m, s_wave[w]);
    __syncthreads();
    return m;
}

// counters: [0] nwork  [1] next work item  [2] scratch cursor
template <int NT>
__global__ __launch_bounds__(NT) void bfs_expand_kernel(
    int N, Thr thr, const int16_t *__restrict__ sem, const int *__restrict__ ball_idx,
    const int *__restrict__ start_len, const int *__restrict__ root, const int *__restrict__ comp_size,
    const int *__restrict__ worklist, int *counters, int *visited, int *claim, int *scratch_node, int *scratch_seed,
    int *cl_size, int *cl_start, int *guard /* strict mode: zeroed [N], else NULL */)
{
    // strict (a graph no ball query of ours vouches for): a list that names one neighbour twice would emit it twice (both
    // edges carry the winning claim; the serial loop skips the second) -- every emission also exchanges a word of `guard`
    // (its own array: the visited flags are read with plain cached loads, which an L2 atomic does not refresh) and a second
    // emission raises counters[17]: the call fails instead of returning a cluster with a repeated member
    const bool strict = guard != nullptr;
    __shared__ int s_pref[NT + 1];
    __shared__ int s_st[NT];
    __shared__ int s_cnt[NT];
    __shared__ int s_wave[NT / 64];
    __shared__ int s_bcast[2];
    // sparse graphs: the frontier (list start / length of its nodes) handed from level to level in LDS, two buffers
    __shared__ int s_fst[2][NT == 512 ? NT : 1], s_fln[2][NT == 512 ? NT : 1];
    // ... and the claims of a level (<= 512 edges) resolved in an LDS hash table instead of global atomics
    constexpr int HS = NT == 512 ? 1024 : 1;
    __shared__ int h_key[HS], h_val[HS];
    const int tid = threadIdx.x;
    constexpr int NW = NT / 64;  // NT = 512 threads for sparse graphs, 1024 for dense / capped ones
    const int nwork = ld_agent(&counters[0]);

    for (;;) {
        if (tid == 0) s_bcast[0] = atomicAdd(&counters[1], 1);
        __syncthreads();
        const int w = s_bcast[0];
        __syncthreads();
        if (w >= nwork) break;
        const int r = worklist[w];
        const int sz = comp_size[r];
        if (tid == 0) s_bcast[1] = atomicAdd(&counters[2], sz);
        __syncthreads();
        int tail = s_bcast[1];
        __syncthreads();
        const int lab = thr.mode == 0 ? (int)sem[r] : 0;
        int done = 0;
        int seed = r;  // the root is the smallest member, hence the first seed
        while (done < sz) {
            if (done > 0) {
                // next seed: smallest unvisited member of this component above the previous seed
                int found = INT_BIG;
                for (int c = seed + 1; c < N && found == INT_BIG; c += NT) {
                    const int i = c + tid;
                    const int cand = (i < N && root[i] == r && visited[i] == 0) ? i : INT_BIG;
                    found = block_min_512<NT>(cand, s_wave);
                }
                seed = found;
                if (seed == INT_BIG) break;  // cannot happen: done < sz guarantees a member is left
            }
            const int cluster_start = tail;
            if (tid == 0) {
                scratch_node[tail] = seed;
                scratch_seed[tail] = seed;
                visited[seed] = 1;
            }
            tail += 1;
            int lvl_begin = cluster_start, lvl_end = tail;
            __syncthreads();
            int fbuf = 0;
            bool f_valid = false;   // s_fst / s_fln [fbuf] hold the current frontier
            while (lvl_begin < lvl_end) {
                int new_tail = tail;
                if (NT == 512 && lvl_end - lvl_begin <= NT) {
                    // ---- sparse graphs, frontier of at most 512 nodes (every level of an object at 3 cm radius): one edge
                    // per thread, its target kept in a register from the claim to the win test, the next frontier's list
                    // headers fetched while the winners are written -- 3-4 dependent memory round trips per level instead
                    // of 8 (a level is nothing but round trips here: ~200 levels per object).  More than 512 edges in
                    // the level: the general path below.
                    const int F = lvl_end - lvl_begin;
                    if (!f_valid) {
                        if (tid < F) {
                            const int node = scratch_node[lvl_begin + tid];
                            s_fst[fbuf][tid] = start_len[node * 2];
                            s_fln[fbuf][tid] = start_len[node * 2 + 1];
                        }
                        __syncthreads();
                    }
                    int E;
                    const int ex = block_excl_scan_512<NT>(tid < F ? s_fln[fbuf][tid] : 0, &E, s_wave);
                    if (E <= NT) {
                        s_pref[tid] = ex;
                        if (tid == 0) s_pref[NT] = E;
                        for (int q = tid; q < HS; q += NT) {
                            h_key[q] = -1;
                            h_val[q] = INT_BIG;
                        }
                        __syncthreads();
                        int j = -1, lo = 0, slot = 0;
                        int2 hdr = make_int2(0, 0);
                        bool ok = false;
                        if (tid < E) {
                            int hi = F;  // largest p with pref[p] <= e
                            while (hi - lo > 1) {
                                const int mid = (lo + hi) >> 1;
                                if (s_pref[mid] <= tid) lo = mid; else hi = mid;
                            }
                            j = ball_idx[s_fst[fbuf][lo] + (tid - s_pref[lo])];
                            // one round trip for everything that hangs on j: label, visited flag AND the target's own list
                            // header (needed only if it wins -- fetched with the test instead of after it: a level is
                            // two dependent round trips, edge targets -> target records, instead of three)
                            const int lj = thr.mode == 0 ? (int)sem[j] : lab;
                            const int vj = visited[j];
                            hdr = *reinterpret_cast<const int2 *>(start_len + 2 * (size_t)j);
                            ok = lj == lab && vj == 0;
                            if (ok) {
                                // all edges of the level are in this workgroup: the smallest parent position per target
                                // is settled in LDS (open addressing, <= 512 keys in 1024 slots), no memory-side atomics
                                slot = (int)(((unsigned)j * 2654435761u) >> 22) & (HS - 1);
                                for (;;) {
                                    const int k = atomicCAS(&h_key[slot], -1, j);
                                    if (k == -1 || k == j) break;
                                    slot = (slot + 1) & (HS - 1);
                                }
                                atomicMin(&h_val[slot], lo);
                            }
                        }
                        int tot = 0;
                        if (__syncthreads_or(ok ? 1 : 0)) {
                            const int win = (ok && h_val[slot] == lo) ? 1 : 0;
                            const int rank = block_excl_scan_512<NT>(win, &tot, s_wave);
                            if (win) {
                                scratch_node[new_tail + rank] = j;
                                scratch_seed[new_tail + rank] = seed;
                                if (strict && atomicExch(&guard[j], 1) == 1) counters[17] = 1;
                                visited[j] = 1;
                                if (rank < NT) {   // (tot <= E <= NT)
                                    s_fst[fbuf ^ 1][rank] = hdr.x;
                                    s_fln[fbuf ^ 1][rank] = hdr.y;
                                }
                            }
                        }
                        __syncthreads();
                        fbuf ^= 1;
                        f_valid = true;
                        new_tail += tot;
                        lvl_begin = lvl_end;
                        lvl_end = new_tail;
                        tail = new_tail;
                        continue;
                    }
                    __syncthreads();
                }
                f_valid = false;
                for (int c0 = lvl_begin; c0 < lvl_end; c0 += NT) {
                    const int cn = min(NT, lvl_end - c0);
                    // dense lists (capped shifted-coordinate graphs: up to 1000 neighbours) go one wave per frontier node;
                    // short lists (raw coordinates: ~15 neighbours) keep the flat edge index, where no lane idles
                    // NT == 1024 is launched for dense graphs (n_edges >= 24 N) only
                    constexpr bool dense_lists = NT == 1024;
                    if (dense_lists) {
                        // one wave per frontier node, lanes stride its neighbour list (the former flat edge index needed a
                        // 9-step bisection per edge and a workgroup scan per 512 edges: 28 ms for 48 capped components)
                        if (tid < cn) {
                            const int node = scratch_node[c0 + tid];
                            s_st[tid] = start_len[node * 2];
                            s_pref[tid] = start_len[node * 2 + 1];   // length for now, winner offset later
                        }
                        __syncthreads();
                        const int wv = tid >> 6, ln_ = tid & 63;
                        // ---- phase A: every out-edge to an unvisited, label-compatible node posts its parent position
                        int any = 0;
                        for (int pos = wv; pos < cn; pos += NW) {
                            const int st = s_st[pos], ln = s_pref[pos];
                            for (int t = ln_; t < ln; t += 64) {
                                const int j = ball_idx[st + t];
                                if (visited[j]) continue;
                                if (thr.mode == 0 && (int)sem[j] != lab) continue;
                                if (ld_agent(&claim[j]) > pos) atomicMin(&claim[j], pos);
                                any = 1;
                            }
                        }
                        any = __syncthreads_or(any);
                        if (any) {
                            // ---- phase B1: winners per node
                            int my_cnt = 0;
                            for (int pos = wv; pos < cn; pos += NW) {
                                const int st = s_st[pos], ln = s_pref[pos];
                                int c = 0;
                                for (int t0 = 0; t0 < ln; t0 += 64) {
                                    const int t = t0 + ln_;
                                    bool win = false;
                                    if (t < ln) {
                                        const int j = ball_idx[st + t];
                                        win = visited[j] == 0 && (thr.mode != 0 || (int)sem[j] == lab) && ld_agent(&claim[j]) == pos;
                                    }
                                    c += __popcll(__ballot(win));
                                }
                                if (ln_ == 0) s_cnt[pos] = c;
                            }
                            __syncthreads();
                            my_cnt = tid < cn ? s_cnt[tid] : 0;
                            int tot;
                            const int off = block_excl_scan_512<NT>(my_cnt, &tot, s_wave);
                            if (tid < cn) s_cnt[tid] = off;
                            __syncthreads();
                            // ---- phase B2: winners compacted in (parent position, slot) order
                            for (int pos = wv; pos < cn; pos += NW) {
                                const int st = s_st[pos], ln = s_pref[pos];
                                int out = new_tail + s_cnt[pos];
                                for (int t0 = 0; t0 < ln; t0 += 64) {
                                    const int t = t0 + ln_;
                                    bool win = false;
                                    int j = -1;
                                    if (t < ln) {
                                        j = ball_idx[st + t];
                                        win = visited[j] == 0 && (thr.mode != 0 || (int)sem[j] == lab) && ld_agent(&claim[j]) == pos;
                                    }
                                    const unsigned long long m = __ballot(win);
                                    if (win) {
                                        const int o = out + ballot_rank(m);
                                        scratch_node[o] = j;
                                        scratch_seed[o] = seed;
                                    }
                                    out += __popcll(m);
                                }
                            }
                            __syncthreads();
                            // mark the winners visited only now: phase B2 of another wave must still see them unvisited
                            for (int o = new_tail + tid; o < new_tail + tot; o += NT) {
                                if (strict && atomicExch(&guard[scratch_node[o]], 1) == 1) counters[17] = 1;
                                visited[scratch_node[o]] = 1;
                            }
                            new_tail += tot;
                        }
                        __syncthreads();
                        continue;
                    }
                    int ln = 0, st = 0;
                    if (tid < cn) {
                        const int node = scratch_node[c0 + tid];
                        st = start_len[node * 2];
                        ln = start_len[node * 2 + 1];
                    }
                    int E;
                    const int ex = block_excl_scan_512<NT>(ln, &E, s_wave);
                    s_pref[tid] = ex;
                    s_st[tid] = st;
                    if (tid == 0) s_pref[NT] = E;
                    __syncthreads();
                    // ---- phase A: every out-edge to an unvisited, label-compatible node posts its parent position
                    int any = 0;
                    for (int e = tid; e < E; e += NT) {
                        int lo = 0, hi = cn;  // largest p with pref[p] <= e
                        while (hi - lo > 1) {
                            const int mid = (lo + hi) >> 1;
                            if (s_pref[mid] <= e) lo = mid; else hi = mid;
                        }
                        const int j = ball_idx[s_st[lo] + (e - s_pref[lo])];
                        if (thr.mode == 0 && (int)sem[j] != lab) continue;
                        if (visited[j]) continue;
                        atomicMin(&claim[j], lo);
                        any = 1;
                    }
                    any = __syncthreads_or(any);
                    if (any) {
                        // ---- phase B: winners, compacted in (parent position, slot) order
                        for (int e0 = 0; e0 < E; e0 += NT) {
                            const int e = e0 + tid;
                            int win = 0, j = -1;
                            if (e < E) {
                                int lo = 0, hi = cn;
                                while (hi - lo > 1) {
                                    const int mid = (lo + hi) >> 1;
                                    if (s_pref[mid] <= e) lo = mid; else hi = mid;
                                }
                                j = ball_idx[s_st[lo] + (e - s_pref[lo])];
                                const bool ok = (thr.mode != 0 || (int)sem[j] == lab) && visited[j] == 0;
                                if (ok && ld_agent(&claim[j]) == lo) win = 1;
                            }
                            int tot;
                            const int rank = block_excl_scan_512<NT>(win, &tot, s_wave);
                            if (win) {
                                scratch_node[new_tail + rank] = j;
                                scratch_seed[new_tail + rank] = seed;
                                if (strict && atomicExch(&guard[j], 1) == 1) counters[17] = 1;
                                visited[j] = 1;
                            }
                            new_tail += tot;
                            __syncthreads();
                        }
                    }
                    __syncthreads();
                }
                lvl_begin = lvl_end;
                lvl_end = new_tail;
                tail = new_tail;
            }
            const int size = tail - cluster_start;
            if (tid == 0) {
                cl_size[seed] = size;
                cl_start[seed] = cluster_start;
            }
            done += size;
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Chip-wide level-synchronous expansion for SYMMETRIC graphs (no list reached the cap): then a weak component is
// exactly one cluster, seeded by its root, and all qualifying components can be expanded together by every CU.
// The frontier array F holds, per level, each component's frontier as one contiguous segment in queue order.
//
// A node j joins level L+1 under the frontier node with the SMALLEST queue position p among its label-compatible
// neighbours, and the children of p are appended in ascending index (= the order of p's list): exactly what the
// serial FIFO loop does.  Posting atomicMin(claim[j], p) per edge -- round 1 -- costs a memory-side atomic per
// attempt (5.7 G/s on this part against 240 G/s for cached gathers, tools/gather_micro.hip; the L2 of an XCD does not
// see the other seven's atomics, so most edges towards an unvisited node attempt one).  Three launches per level
// without hot atomics instead, every per-edge access a plain 8-byte gather of the node word
//     vp[i] = level:16 | label:16 | frontier position:32        (level 0xFFFF = unvisited):
//   mark : frontier node p takes its final queue slot and stores cand[j] = L+1 on every unvisited, label-compatible
//          neighbour (plain store, idempotent).  The symmetric graph turns the question around:
//   pull : every candidate scans ITS OWN list for the smallest position among the level-L nodes -- that is its parent
//          p* -- and leaves cc[j] = (L+1, p*), cnt[p*] += 1 (one atomic per discovered node).
//   win  : single-pass ordered compaction: tiles of frontier positions off a ticket counter, exclusive scan of cnt[]
//          inside the tile, tile totals chained by decoupled look-back; the positions with children scan their list for
//          cc[j] == (L+1, p) and write the children to F_next in list order (vp[j] = visited at level L+1).
// The per-component bookkeeping of the next level (queue base, segment start / size: double-buffered by level parity)
// is advanced in `mark`.
// counters: [6] |F| of even levels  [7] |F| of odd levels  [8] tile ticket
constexpr int WIN_TILE = 64;          // frontier positions per tile
constexpr unsigned long long VP_UNVISITED = 0xFFFFull << 48;

__device__ __forceinline__ unsigned long long vp_make(int level, int label, int pos)
{
    return ((unsigned long long)(unsigned)(level & 0xFFFF) << 48) | ((unsigned long long)(unsigned)(label & 0xFFFF) << 32) |
           (unsigned long long)(unsigned)pos;
}

__global__ void glob_vp_init_kernel(int N, Thr thr, const int16_t *__restrict__ sem, unsigned long long *vp, int *cand,
                                    unsigned long long *cc)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    vp[i] = vp_make(0xFFFF, thr.mode == 0 ? (int)sem[i] : 0, 0);
    cand[i] = 0;
    cc[i] = 0ull;  // a claim left behind by an earlier call on this workspace must not match a (level, position) of this one
}

__global__ void glob_init_kernel(const int *__restrict__ worklist, const int *__restrict__ comp_size, int *counters,
                                 int *F0, int *comp_base, int *done, int *seg_start, int *seg_cnt, unsigned long long *vp)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    const int nwork = counters[0];
    if (w == 0) counters[6] = nwork;
    if (w >= nwork) return;  // launched for the upper bound N
    const int r = worklist[w];
    F0[w] = r;
    comp_base[r] = atomicAdd(&counters[2], comp_size[r]);
    done[r] = 0;
    seg_start[r] = w;
    seg_cnt[r] = 1;
    vp[r] = (vp[r] & (0xFFFFull << 32)) | (unsigned long long)(unsigned)w;  // the seed: level 0, position w
}

__global__ __launch_bounds__(256) void glob_mark_kernel(
    int level, const int *__restrict__ ball_idx, const int *__restrict__ start_len, const int *__restrict__ root,
    const int *__restrict__ F, int *counters, const int *__restrict__ worklist, const int *__restrict__ comp_base,
    const int *__restrict__ done_cur, int *done_next, const int *__restrict__ seg_start_cur, int *seg_start_next,
    const int *__restrict__ seg_cnt_cur, int *seg_cnt_next, const unsigned long long *__restrict__ vp, int *cand, int *cnt,
    int *scratch_node, int *scratch_seed, unsigned long long *tile_status)
{
    const int nF = counters[6 + (level & 1)];
    const int gid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    {
        const int nwork = counters[0];
        for (int w = gid; w < nwork; w += gsz) {
            const int r = worklist[w];
            done_next[r] = done_cur[r] + seg_cnt_cur[r];
            seg_start_next[r] = INT_BIG;
            seg_cnt_next[r] = 0;
        }
        const int ntiles = (nF + WIN_TILE - 1) / WIN_TILE;
        for (int t = gid; t < ntiles; t += gsz) tile_status[t] = 0ull;
        if (gid == 0) {
            counters[8] = 0;
            counters[6 + ((level + 1) & 1)] = 0;
            if (nF > 0) counters[14] = level + 1;   // depth so far: the host sizes the next call's speculative launch by it
        }
    }
    const int waves = blockDim.x >> 6, l = lane_id();
    for (int p = blockIdx.x * waves + wave_id(); p < nF; p += gridDim.x * waves) {
        const int node = F[p];
        const int r = root[node];
        if (l == 0) {
            const int qpos = comp_base[r] + done_cur[r] + (p - seg_start_cur[r]);
            scratch_node[qpos] = node;
            scratch_seed[qpos] = r;
            cnt[p] = 0;
        }
        const int st = start_len[node * 2], ln = start_len[node * 2 + 1];
        const unsigned long long want = VP_UNVISITED | (vp[node] & (0xFFFFull << 32));  // unvisited, my label
        // four 64-edge slices per trip: the four index loads, then the four node-word gathers are in flight together
        for (int t0 = 0; t0 < ln; t0 += 256) {
            int j[4];
            unsigned long long v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int t = t0 + 64 * u + l;
                j[u] = t < ln ? ball_idx[st + t] : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = j[u] >= 0 ? vp[j[u]] : 0ull;
#pragma unroll
            for (int u = 0; u < 4; u++)
                if ((v[u] >> 32) == (want >> 32)) cand[j[u]] = level + 1;
        }
    }
}

// one workgroup per 256 nodes: the candidates among them are compacted in LDS, then its waves take them in turn
__global__ __launch_bounds__(256) void glob_pull_kernel(int N, int level, const int *__restrict__ ball_idx,
                                                        const int *__restrict__ start_len,
                                                        const unsigned long long *__restrict__ vp,
                                                        const int *__restrict__ cand, unsigned long long *__restrict__ cc,
                                                        int *cnt)
{
    __shared__ int s_list[256];
    __shared__ int s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool is_cand = i < N && cand[i] == level + 1;
    const unsigned long long m = __ballot(is_cand);
    int base = 0;
    if (lane_id() == 0 && m) base = atomicAdd(&s_n, __popcll(m));
    base = __shfl(base, 0, 64);
    if (is_cand) s_list[base + ballot_rank(m)] = i;
    __syncthreads();
    const int n = s_n, l = lane_id();
    for (int k = wave_id(); k < n; k += 4) {
        const int node = s_list[k];
        const int st = start_len[node * 2], ln = start_len[node * 2 + 1];
        const unsigned long long want = vp_make(level, (int)((vp[node] >> 32) & 0xFFFF), 0) >> 32;  // level L, my label
        unsigned best = 0xFFFFFFFFu;
        for (int t0 = 0; t0 < ln; t0 += 256) {
            int j[4];
            unsigned long long v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int t = t0 + 64 * u + l;
                j[u] = t < ln ? ball_idx[st + t] : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = j[u] >= 0 ? vp[j[u]] : ~0ull;
#pragma unroll
            for (int u = 0; u < 4; u++)
                if ((v[u] >> 32) == want) best = min(best, (unsigned)v[u]);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) best = min(best, (unsigned)__shfl_xor((int)best, d, 64));
        if (l == 0 && best != 0xFFFFFFFFu) {
            cc[node] = ((unsigned long long)(unsigned)(level + 1) << 32) | best;
            atomicAdd(&cnt[best], 1);
        }
    }
}

__global__ __launch_bounds__(256) void glob_win_kernel(int level, const int *__restrict__ ball_idx,
                                                       const int *__restrict__ start_len,
                                                       const int *__restrict__ root, const int *__restrict__ F,
                                                       int *counters, const int *__restrict__ cnt,
                                                       const unsigned long long *__restrict__ cc, unsigned long long *vp,
                                                       unsigned long long *tile_status, int *F_next, int *seg_start_next,
                                                       int *seg_cnt_next)
{
    __shared__ int s_cnt[WIN_TILE];
    __shared__ int s_off[WIN_TILE];
    __shared__ int s_live[WIN_TILE];   // positions of the tile that have children
    __shared__ int s_bcast[3];
    const int nF = counters[6 + (level & 1)];
    const int ntiles = (nF + WIN_TILE - 1) / WIN_TILE;
    const int l = lane_id(), wv = wave_id();
    constexpr unsigned long long FLAG_AGG = 1ull << 62, FLAG_INCL = 2ull << 62, VAL = (1ull << 62) - 1;
    if ((int)blockIdx.x >= ntiles) return;  // surplus workgroups leave without touching the ticket counter
    for (;;) {
        if (threadIdx.x == 0) s_bcast[0] = atomicAdd(&counters[8], 1);
        __syncthreads();
        const int tile = s_bcast[0];
        if (tile >= ntiles) break;
        // ---- tile scan of the child counts + decoupled look-back over the preceding tiles (wave 0)
        if (wv == 0) {
            const int pos = tile * WIN_TILE + l;
            const int v = pos < nF ? cnt[pos] : 0;
            const int incl = wave_incl_scan(v);
            s_cnt[l] = v;
            s_off[l] = incl - v;
            const unsigned long long live = __ballot(v > 0);
            if (v > 0) s_live[ballot_rank(live)] = l;
            const int total = __shfl(incl, 63, 64);
            // relaxed on purpose: the word carries its own payload, nothing else is ordered by it (an acquire / release
            // pair at agent scope invalidates / writes back the L2 on every poll: a 10x slower kernel)
            if (l == 0)
                __hip_atomic_store(&tile_status[tile], (tile == 0 ? FLAG_INCL : FLAG_AGG) | (unsigned long long)total,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // look-back, 64 predecessors per step: sum the aggregates down to the nearest tile whose inclusive prefix
            // is known (tile -1 counts as "inclusive 0")
            int base = 0;
            for (int hi = tile - 1; hi >= 0;) {
                const int t = hi - l;
                const unsigned long long sv = t >= 0 ? __hip_atomic_load(&tile_status[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                     : FLAG_INCL;
                const unsigned long long incl_m = __ballot((sv >> 62) == 2ull), inval_m = __ballot((sv >> 62) == 0ull);
                const int first = incl_m ? __ffsll((long long)incl_m) - 1 : 64;
                const unsigned long long need = first >= 63 ? ~0ull : ((2ull << first) - 1ull);
                if (inval_m & need) {
                    __builtin_amdgcn_s_sleep(2);  // a predecessor in the window has not published yet: poll again
                    continue;
                }
                base += wave_sum(l <= first ? (int)(sv & VAL) : 0);
                if (first < 64) break;
                hi -= 64;
            }
            if (l == 0) {
                if (tile > 0)
                    __hip_atomic_store(&tile_status[tile], FLAG_INCL | (unsigned long long)(base + total), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                s_bcast[1] = base;
                s_bcast[2] = __popcll(live);
                if (tile == ntiles - 1) counters[6 + ((level + 1) & 1)] = base + total;
            }
        }
        __syncthreads();
        // ---- children out, in (position, slot) order: the waves take the positions that have children in turn
        const int base = s_bcast[1], nlive = s_bcast[2];
        for (int k = wv; k < nlive; k += 4) {
            const int q = s_live[k], pos = tile * WIN_TILE + q;
            const int node = F[pos];
            const int st = start_len[node * 2], ln = start_len[node * 2 + 1];
            int out = base + s_off[q];
            if (l == 0) {
                const int r = root[node];
                atomicMin(&seg_start_next[r], out);
                atomicAdd(&seg_cnt_next[r], s_cnt[q]);
            }
            const unsigned long long mine = ((unsigned long long)(unsigned)(level + 1) << 32) | (unsigned)pos;
            const unsigned long long lab = vp[node] & (0xFFFFull << 32);
            for (int t0 = 0; t0 < ln; t0 += 256) {
                int j[4];
                unsigned long long c[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int t = t0 + 64 * u + l;
                    j[u] = t < ln ? ball_idx[st + t] : -1;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) c[u] = j[u] >= 0 ? cc[j[u]] : 0ull;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const bool win = c[u] == mine;
                    const unsigned long long wm = __ballot(win);
                    if (win) {
                        const int o = out + ballot_rank(wm);
                        F_next[o] = j[u];
                        vp[j[u]] = ((unsigned long long)(unsigned)((level + 1) & 0xFFFF) << 48) | lab | (unsigned)o;
                    }
                    out += __popcll(wm);
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// The same three passes with LDS-resident node BITMAPS (round 5).  The passes above look every edge's target up with an
// 8-byte gather of its node word: ~3 x 120 M gathers per PointGroup graph against ~240 G cached gathers / s on this part
// (tools/gather_micro.hip) -- that, not the 480 MB edge list, is what a level costs.  But each pass asks ONE yes / no
// question per edge before it needs anything else:
//   mark : is j unvisited?                       -> bitmap V  (bit j = node j has a level)
//   pull : is j in the frontier of level L?      -> bitmap F  (bit j = level(j) == L), then the node word of those only
//   win  : was j claimed for level L + 1?        -> bitmap C  (set by pull), then the claim word of those only
// N bits are 50 KB for the 400 k foreground points of a 4-scene batch: every workgroup copies the level's bitmap into LDS
// (1024 threads, two workgroups per CU: 25 MB of L2 reads per pass) and an edge costs one ds_read instead of one L2
// gather; the 8-byte gathers that remain are the frontier members (pull) and the claimed children (win).
// `mark` no longer tests the label of j (it has only the bit): a neighbour of another label becomes a candidate, scans
// its own list in `pull` for level-L nodes of ITS label, and either finds none (nothing happens) or finds its true
// parent -- it was a legitimate candidate of its own component then, whose frontier marks it as well.  Same clusters,
// same order (tests: golden / KAT / full-scene cases run through these kernels by default; MS3D_BFS_BITMAP=0 = the
// passes above).  glob_bits_kernel builds V and F from the node words and clears C once per level.
constexpr int BM_THREADS = 1024;

__global__ void glob_bits_kernel(int N, int level, const unsigned long long *__restrict__ vp, unsigned long long *Vb,
                                 unsigned long long *Fb, unsigned long long *Cb)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned lvl = i < N ? (unsigned)(vp[i] >> 48) : 0xFFFFu;
    const unsigned long long mv = __ballot(lvl != 0xFFFFu), mf = __ballot(i < N && lvl == (unsigned)(level & 0xFFFF));
    if (lane_id() == 0 && (i >> 6) <= ((N - 1) >> 6)) {
        Vb[i >> 6] = mv;
        Fb[i >> 6] = mf;
        Cb[i >> 6] = 0ull;
    }
}

__device__ __forceinline__ void bm_load(unsigned *s_bits, const unsigned long long *__restrict__ bm, int nwords64)
{
    // 16 bytes per lane; the bitmap arrays are 16-byte aligned and padded to an even number of 64-bit words
    const uint4 *src = reinterpret_cast<const uint4 *>(bm);
    uint4 *dst = reinterpret_cast<uint4 *>(s_bits);
    for (int w = threadIdx.x; w < (nwords64 + 1) / 2; w += blockDim.x) dst[w] = src[w];
}
__device__ __forceinline__ bool bm_test(const unsigned *s_bits, int j) { return (s_bits[j >> 5] >> (j & 31)) & 1u; }

__global__ __launch_bounds__(BM_THREADS) void glob_mark_bm_kernel(
    int level, const int *__restrict__ ball_idx, const int *__restrict__ start_len, const int *__restrict__ root,
    const int *__restrict__ F, int *counters, const int *__restrict__ worklist, const int *__restrict__ comp_base,
    const int *__restrict__ done_cur, int *done_next, const int *__restrict__ seg_start_cur, int *seg_start_next,
    const int *__restrict__ seg_cnt_cur, int *seg_cnt_next, int *cand, int *cnt, int *scratch_node, int *scratch_seed,
    unsigned long long *tile_status, const unsigned long long *__restrict__ Vb, int nwords64)
{
    extern __shared__ unsigned s_bits[];
    const int nF = counters[6 + (level & 1)];
    const int gid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    {
        const int nwork = counters[0];
        for (int w = gid; w < nwork; w += gsz) {
            const int r = worklist[w];
            done_next[r] = done_cur[r] + seg_cnt_cur[r];
            seg_start_next[r] = INT_BIG;
            seg_cnt_next[r] = 0;
        }
        const int ntiles = (nF + WIN_TILE - 1) / WIN_TILE;
        for (int t = gid; t < ntiles; t += gsz) tile_status[t] = 0ull;
        if (gid == 0) {
            counters[8] = 0;
            counters[6 + ((level + 1) & 1)] = 0;
            if (nF > 0) counters[14] = level + 1;
        }
    }
    const int waves = blockDim.x >> 6, l = lane_id();
    if (blockIdx.x * waves >= nF) return;            // an exhausted level (or a surplus workgroup): no bitmap copy
    bm_load(s_bits, Vb, nwords64);
    __syncthreads();
    // A frontier node is a chain of dependent round trips -- F[p] -> list header / root -> list slices / component
    // bookkeeping -- in front of a few LDS bit tests: the wave's nodes go through a three-stage software pipeline
    // (node A's edges are tested while node B's slices, node C's header and node D's id are in flight).
    const int stride = gridDim.x * waves;
    struct Hdr { int node, st, ln, r; };
    struct Body { int j[4], base, done, seg; };
    auto load_id = [&](int p) { return p < nF ? F[p] : -1; };
    auto load_hdr = [&](int node) {
        Hdr h{node, 0, 0, 0};
        if (node >= 0) { h.st = start_len[node * 2]; h.ln = start_len[node * 2 + 1]; h.r = root[node]; }
        return h;
    };
    auto load_body = [&](const Hdr &h) {
        Body b;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int t = 64 * u + l;
            b.j[u] = (h.node >= 0 && t < h.ln) ? ball_idx[h.st + t] : -1;
        }
        b.base = b.done = b.seg = 0;
        if (h.node >= 0) { b.base = comp_base[h.r]; b.done = done_cur[h.r]; b.seg = seg_start_cur[h.r]; }
        return b;
    };
    int p = blockIdx.x * waves + wave_id();
    Hdr hA = load_hdr(load_id(p)), hB = load_hdr(load_id(p + stride));
    int idC = load_id(p + 2 * stride);
    Body bA = load_body(hA);
    for (; p < nF; p += stride) {
        const int idD = load_id(p + 3 * stride);
        const Hdr hC = load_hdr(idC);
        const Body bB = load_body(hB);
        // ---- node A
        if (l == 0) {
            const int qpos = bA.base + bA.done + (p - bA.seg);
            scratch_node[qpos] = hA.node;
            scratch_seed[qpos] = hA.r;
            cnt[p] = 0;
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (bA.j[u] >= 0 && !bm_test(s_bits, bA.j[u])) cand[bA.j[u]] = level + 1;
        for (int t0 = 256; t0 < hA.ln; t0 += 256) {          // lists beyond 256 entries: the rest, slice group by slice group
            int j[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int t = t0 + 64 * u + l;
                j[u] = t < hA.ln ? ball_idx[hA.st + t] : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (j[u] >= 0 && !bm_test(s_bits, j[u])) cand[j[u]] = level + 1;
        }
        hA = hB; bA = bB; hB = hC; idC = idD;
    }
}

// a workgroup per 1024 nodes: the candidates among them are compacted in LDS, then its 16 waves take them in turn
__global__ __launch_bounds__(BM_THREADS) void glob_pull_bm_kernel(int N, int level, const int *__restrict__ ball_idx,
                                                                  const int *__restrict__ start_len,
                                                                  const unsigned long long *__restrict__ vp,
                                                                  const int *__restrict__ cand,
                                                                  unsigned long long *__restrict__ cc, int *cnt,
                                                                  const int *__restrict__ counters,
                                                                  const unsigned long long *__restrict__ Fb, unsigned *Cb32,
                                                                  int nwords64)
{
    extern __shared__ unsigned s_bits[];
    __shared__ int s_list[BM_THREADS];
    __shared__ int s_n;
    if (counters[6 + (level & 1)] == 0) return;      // exhausted level
    bm_load(s_bits, Fb, nwords64);
    const int l = lane_id(), waves = blockDim.x >> 6;
    for (int base_node = blockIdx.x * BM_THREADS; base_node < N; base_node += gridDim.x * BM_THREADS) {
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
        const int i = base_node + threadIdx.x;
        const bool is_cand = i < N && cand[i] == level + 1;
        const unsigned long long m = __ballot(is_cand);
        int base = 0;
        if (l == 0 && m) base = atomicAdd(&s_n, __popcll(m));
        base = __shfl(base, 0, 64);
        if (is_cand) s_list[base + ballot_rank(m)] = i;
        __syncthreads();
        const int n = s_n;
        // two-stage pipeline over the wave's candidates: node B's header / own word, then its first slices, are requested
        // while node A's slices are tested (the candidate ids come from LDS)
        struct Hdr { int node, st, ln; unsigned long long w; };
        auto load_hdr = [&](int k) {
            Hdr h{-1, 0, 0, 0ull};
            if (k < n) { h.node = s_list[k]; h.st = start_len[h.node * 2]; h.ln = start_len[h.node * 2 + 1]; h.w = vp[h.node]; }
            return h;
        };
        auto load_slices = [&](const Hdr &h, int (&j)[4]) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int t = 64 * u + l;
                j[u] = (h.node >= 0 && t < h.ln) ? ball_idx[h.st + t] : -1;
            }
        };
        int k = wave_id();
        Hdr hA = load_hdr(k), hB = load_hdr(k + waves);
        int jA[4];
        load_slices(hA, jA);
        for (; k < n; k += waves) {
            const Hdr hC = load_hdr(k + 2 * waves);
            int jB[4];
            load_slices(hB, jB);
            const unsigned long long want = vp_make(level, (int)((hA.w >> 32) & 0xFFFF), 0) >> 32;  // level L, my label
            unsigned best = 0xFFFFFFFFu;
            {
                unsigned long long v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) v[u] = (jA[u] >= 0 && bm_test(s_bits, jA[u])) ? vp[jA[u]] : ~0ull;
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if ((v[u] >> 32) == want) best = min(best, (unsigned)v[u]);
            }
            for (int t0 = 256; t0 < hA.ln; t0 += 256) {
                int j[4];
                unsigned long long v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int t = t0 + 64 * u + l;
                    j[u] = t < hA.ln ? ball_idx[hA.st + t] : -1;
                }
#pragma unroll
                for (int u = 0; u < 4; u++) v[u] = (j[u] >= 0 && bm_test(s_bits, j[u])) ? vp[j[u]] : ~0ull;
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if ((v[u] >> 32) == want) best = min(best, (unsigned)v[u]);
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) best = min(best, (unsigned)__shfl_xor((int)best, d, 64));
            if (l == 0 && best != 0xFFFFFFFFu) {
                cc[hA.node] = ((unsigned long long)(unsigned)(level + 1) << 32) | best;
                atomicAdd(&cnt[best], 1);
                atomicOr(&Cb32[hA.node >> 5], 1u << (hA.node & 31));
            }
            hA = hB; hB = hC;
#pragma unroll
            for (int u = 0; u < 4; u++) jA[u] = jB[u];
        }
        __syncthreads();
    }
}

// win with bitmaps: a workgroup takes tiles of WIN_BM_TILE frontier positions off the ticket counter, compacts the
// positions that have children, with their output offsets, into an LDS list, and its waves walk that list through the
// same three-stage pipeline as `mark` (id -> list header -> list slices in flight for three entries at once).
// (tile size: a level's frontier is 10^4..10^5 positions and every tile is one workgroup's work item -- 1024-position
// tiles left 200 of 256 CUs without one; 128 positions x 512 threads, two workgroups per CU, measured best)
constexpr int WIN_BM_THREADS = 512;
constexpr int WIN_BM_TILE = 128;
static_assert(WIN_BM_TILE % WIN_TILE == 0, "tile_status is cleared per WIN_TILE positions: a coarser tile uses a prefix of it");

__global__ __launch_bounds__(WIN_BM_THREADS) void glob_win_bm_kernel(int level, const int *__restrict__ ball_idx,
                                                                 const int *__restrict__ start_len,
                                                                 const int *__restrict__ root, const int *__restrict__ F,
                                                                 int *counters, const int *__restrict__ cnt,
                                                                 const unsigned long long *__restrict__ cc,
                                                                 unsigned long long *vp, unsigned long long *tile_status,
                                                                 int *F_next, int *seg_start_next, int *seg_cnt_next,
                                                                 const unsigned long long *__restrict__ Cb, int nwords64)
{
    extern __shared__ unsigned s_bits[];
    __shared__ int s_wtot[WIN_BM_THREADS / 64];
    __shared__ int s_wbase[WIN_BM_THREADS / 64];
    __shared__ int s_lpos[WIN_BM_TILE];    // live entries: position in the tile | children << 16 ... kept as two arrays
    __shared__ int s_lout[WIN_BM_TILE];
    __shared__ int s_lcnt[WIN_BM_TILE];
    __shared__ int s_bcast[3];
    const int nF = counters[6 + (level & 1)];
    const int ntiles = (nF + WIN_BM_TILE - 1) / WIN_BM_TILE;
    const int l = lane_id(), wv = wave_id(), waves = blockDim.x >> 6;
    constexpr unsigned long long FLAG_AGG = 1ull << 62, FLAG_INCL = 2ull << 62, VAL = (1ull << 62) - 1;
    if ((int)blockIdx.x >= ntiles) return;  // surplus workgroups leave without touching the ticket counter
    bm_load(s_bits, Cb, nwords64);
    for (;;) {
        if (threadIdx.x == 0) { s_bcast[0] = atomicAdd(&counters[8], 1); s_bcast[2] = 0; }
        __syncthreads();
        const int tile = s_bcast[0];
        if (tile >= ntiles) break;
        const int pos = tile * WIN_BM_TILE + threadIdx.x;
        const int v = ((int)threadIdx.x < WIN_BM_TILE && pos < nF) ? cnt[pos] : 0;
        const int incl = wave_incl_scan(v);
        if (l == 63) s_wtot[wv] = incl;
        __syncthreads();
        if (wv == 0) {
            const int wt = l < waves ? s_wtot[l] : 0;
            const int wincl = wave_incl_scan(wt);
            if (l < waves) s_wbase[l] = wincl - wt;
            const int total = __shfl(wincl, 63, 64);
            // relaxed on purpose (see glob_win_kernel): the word carries its own payload
            if (l == 0)
                __hip_atomic_store(&tile_status[tile], (tile == 0 ? FLAG_INCL : FLAG_AGG) | (unsigned long long)total,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int base = 0;
            for (int hi = tile - 1; hi >= 0;) {
                const int t = hi - l;
                const unsigned long long sv = t >= 0 ? __hip_atomic_load(&tile_status[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                     : FLAG_INCL;
                const unsigned long long incl_m = __ballot((sv >> 62) == 2ull), inval_m = __ballot((sv >> 62) == 0ull);
                const int first = incl_m ? __ffsll((long long)incl_m) - 1 : 64;
                const unsigned long long need = first >= 63 ? ~0ull : ((2ull << first) - 1ull);
                if (inval_m & need) {
                    __builtin_amdgcn_s_sleep(2);
                    continue;
                }
                base += wave_sum(l <= first ? (int)(sv & VAL) : 0);
                if (first < 64) break;
                hi -= 64;
            }
            if (l == 0) {
                if (tile > 0)
                    __hip_atomic_store(&tile_status[tile], FLAG_INCL | (unsigned long long)(base + total), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                s_bcast[1] = base;
                if (tile == ntiles - 1) counters[6 + ((level + 1) & 1)] = base + total;
            }
        }
        __syncthreads();
        {
            // the positions with children -> the live list (any order: an entry carries its own output offset)
            const int out = s_bcast[1] + s_wbase[wv] + incl - v;
            const unsigned long long live = __ballot(v > 0);
            int lb = 0;
            if (l == 0 && live) lb = atomicAdd(&s_bcast[2], __popcll(live));
            lb = __shfl(lb, 0, 64);
            if (v > 0) {
                const int e = lb + ballot_rank(live);
                s_lpos[e] = pos;
                s_lout[e] = out;
                s_lcnt[e] = v;
            }
        }
        __syncthreads();
        const int nlive = s_bcast[2];
        // ---- children out, in (position, slot) order; entries of a wave pipelined: A walked, B's slices, C's header, D's id in flight
        struct Hdr { int pos, out, n, node, st, ln, r; unsigned long long w; };
        auto load_id = [&](int k) {
            Hdr h{0, 0, 0, -1, 0, 0, 0, 0ull};
            if (k < nlive) { h.pos = s_lpos[k]; h.out = s_lout[k]; h.n = s_lcnt[k]; h.node = F[h.pos]; }
            return h;
        };
        auto load_hdr = [&](Hdr &h) {
            if (h.node >= 0) {
                h.st = start_len[h.node * 2]; h.ln = start_len[h.node * 2 + 1]; h.r = root[h.node]; h.w = vp[h.node];
            }
        };
        auto load_slices = [&](const Hdr &h, int (&j)[4]) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int t = 64 * u + l;
                j[u] = (h.node >= 0 && t < h.ln) ? ball_idx[h.st + t] : -1;
            }
        };
        int k = wv;
        Hdr hA = load_id(k), hB = load_id(k + waves), hC = load_id(k + 2 * waves);
        load_hdr(hA); load_hdr(hB);
        int jA[4];
        load_slices(hA, jA);
        for (; k < nlive; k += waves) {
            Hdr hD = load_id(k + 3 * waves);
            load_hdr(hC);
            int jB[4];
            load_slices(hB, jB);
            int out = hA.out;
            if (l == 0) {
                atomicMin(&seg_start_next[hA.r], out);
                atomicAdd(&seg_cnt_next[hA.r], hA.n);
            }
            const unsigned long long mine = ((unsigned long long)(unsigned)(level + 1) << 32) | (unsigned)hA.pos;
            const unsigned long long lab = hA.w & (0xFFFFull << 32);
            for (int t0 = 0; t0 < hA.ln; t0 += 256) {
                int j[4];
                unsigned long long c[4];
                if (t0 == 0) {
#pragma unroll
                    for (int u = 0; u < 4; u++) j[u] = jA[u];
                } else {
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int t = t0 + 64 * u + l;
                        j[u] = t < hA.ln ? ball_idx[hA.st + t] : -1;
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; u++) c[u] = (j[u] >= 0 && bm_test(s_bits, j[u])) ? cc[j[u]] : 0ull;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const bool win = c[u] == mine;
                    const unsigned long long wm = __ballot(win);
                    if (win) {
                        const int o = out + ballot_rank(wm);
                        F_next[o] = j[u];
                        vp[j[u]] = ((unsigned long long)(unsigned)((level + 1) & 0xFFFF) << 48) | lab | (unsigned)o;
                    }
                    out += __popcll(wm);
                }
            }
            hA = hB; hB = hC; hC = hD;
#pragma unroll
            for (int u = 0; u < 4; u++) jA[u] = jB[u];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// DIRECTED dense graphs (some list was cut at the 1000-neighbour cap, e.g. SoftGroup's r = 4 cm grouping): a weak
// component may hold several clusters, found one after the other by the serial algorithm (seeds in ascending index
// order, each cluster = what its seed reaches among the still unvisited points).  Restated without the order: a point
// belongs to the cluster of its SMALLEST ANCESTOR (the smallest index that reaches it along directed edges, itself
// included) -- that ancestor cannot have been taken by an earlier seed (the earlier seed would be a smaller ancestor),
// so it becomes a seed, and no point on the path from it is taken earlier for the same reason.  Hence two stages, both
// chip-wide:
//   stage 1  the weak components (union-find) expanded from their roots, all at once.  The root IS the smallest
//            ancestor of everything it reaches, so what the expansion reaches is a final cluster; on the benchmark's
//            graphs most components are exhausted by it.
//   stage 2  the points left over in the other components (a few hundred per blob: those that are in nobody's
//            1000-lowest list): smallest-ancestor labels by min-propagation along the out-edges among the leftovers
//            (their ancestors are leftovers too: anything reachable from a visited point was visited), cluster sizes by
//            label, and a second expansion from every label that qualifies, following only same-label edges.
// Round 1 replayed every capped graph with one workgroup per component (12.6 ms per SoftGroup step); replaying only the
// components stage 1 does not exhaust still took 8 ms, because the replay pays a seed search per leftover singleton.
// A directed graph cannot be pulled (a node does not know who points at it), so the claims are pushed:
//   claim : frontier node p posts atomicMin(claim[j], p) on its unvisited, label-compatible out-neighbours (behind a
//           plain read and a coherent re-read: the atomics are what costs) and keeps the posted edges as one 64-bit mask
//           per 64-edge slice;
//   win   : single-pass compaction of the edges that won (claim[j] == p) in (p, slot) order, as in the symmetric case.
constexpr int DIR_TILE = 16;          // frontier positions per tile of the win kernel
constexpr int MAX_SLICES = 16;        // 64-edge slices per list (1000-entry cap)

__global__ void dir_init_kernel(const int *__restrict__ worklist, const int *__restrict__ comp_size, int *counters,
                                int *F0, int *comp_base, int *done, int *seg_start, int *seg_cnt, int *claim)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    const int nwork = counters[0];
    if (w == 0) counters[6] = nwork;
    if (w >= nwork) return;  // launched for the upper bound N
    const int r = worklist[w];
    F0[w] = r;
    comp_base[r] = atomicAdd(&counters[2], comp_size[r]);
    done[r] = 0;
    seg_start[r] = w;
    seg_cnt[r] = 1;
    claim[r] = -1;  // the seed is visited
}

// first mask slot of a node's list: strictly increasing with the node, never shared between two nodes
// (canonical lists: start = exclusive prefix sum of the lengths in point order; a graph nobody vouches for -- strict -- may
// order its lists anyhow, e.g. the reference's own ball query hands out starts in atomic order: a fixed stride per node then)
__device__ __forceinline__ long mask_slot(int st, int node, int strict = 0)
{
    return strict ? (long)node * MAX_SLICES : (long)(st >> 6) + node;
}

template <bool GROUPS>   // stage 2: an edge counts only inside one label group (root[] holds the labels then)
__global__ __launch_bounds__(256) void dir_claim_kernel(
    Thr thr, int level, const int16_t *__restrict__ sem, const int *__restrict__ ball_idx,
    const int *__restrict__ start_len, const int *__restrict__ root, const int *__restrict__ F, int *counters,
    const int *__restrict__ worklist, const int *__restrict__ comp_base, const int *__restrict__ done_cur, int *done_next,
    const int *__restrict__ seg_start_cur, int *seg_start_next, const int *__restrict__ seg_cnt_cur, int *seg_cnt_next,
    int *claim, int *scratch_node, int *scratch_seed, unsigned long long *__restrict__ amask,
    unsigned long long *tile_status, int strict)
{
    const int nF = counters[6 + (level & 1)];
    const int gid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    {
        const int nwork = counters[0];
        for (int w = gid; w < nwork; w += gsz) {
            const int r = worklist[w];
            done_next[r] = done_cur[r] + seg_cnt_cur[r];
            seg_start_next[r] = INT_BIG;
            seg_cnt_next[r] = 0;
        }
        const int ntiles = (nF + DIR_TILE - 1) / DIR_TILE;
        for (int t = gid; t < ntiles; t += gsz) tile_status[t] = 0ull;
        if (gid == 0) {
            counters[8] = 0;
            counters[6 + ((level + 1) & 1)] = 0;
        }
    }
    const int waves = blockDim.x >> 6, l = lane_id();
    // the wave's frontier nodes through a three-stage software pipeline (round 5, as glob_mark_bm_kernel: node A's claims are
    // posted while node B's first slices, node C's header / root / label and node D's id are in flight)
    const int stride = gridDim.x * waves;
    struct Hdr { int node, st, ln, r, lab; };
    auto load_id = [&](int p) { return p < nF ? F[p] : -1; };
    auto load_hdr = [&](int node) {
        Hdr h{node, 0, 0, 0, 0};
        if (node >= 0) {
            h.st = start_len[node * 2]; h.ln = min(start_len[node * 2 + 1], 64 * MAX_SLICES);   // canonical lists: <= 1000
            h.r = root[node];
            h.lab = thr.mode == 0 ? (int)sem[node] : 0;
        }
        return h;
    };
    struct Body { int j[4], base, done, seg; };
    auto load_body = [&](const Hdr &h) {
        Body b;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int t = 64 * u + l;
            b.j[u] = (h.node >= 0 && t < h.ln) ? ball_idx[h.st + t] : -1;
        }
        b.base = b.done = b.seg = 0;
        if (h.node >= 0) { b.base = comp_base[h.r]; b.done = done_cur[h.r]; b.seg = seg_start_cur[h.r]; }
        return b;
    };
    int p = blockIdx.x * waves + wave_id();
    Hdr hA = load_hdr(load_id(p)), hB = load_hdr(load_id(p + stride));
    int idC = load_id(p + 2 * stride);
    Body bA = load_body(hA);
    for (; p < nF; p += stride) {
        const int idD = load_id(p + 3 * stride);
        const Hdr hC = load_hdr(idC);
        const Body bB = load_body(hB);
        const int node = hA.node, r = hA.r;
        if (l == 0) {
            const int qpos = bA.base + bA.done + (p - bA.seg);
            scratch_node[qpos] = node;
            scratch_seed[qpos] = r;
        }
        const int st = hA.st, ln = hA.ln;
        const int lab = hA.lab;
        unsigned long long *am = amask + mask_slot(st, node, strict);
        // four 64-edge slices per trip: the four index loads, then the four claim gathers are in flight together
        for (int t0 = 0; t0 < ln; t0 += 256) {
            int j[4], c[4];
            if (t0 == 0) {
#pragma unroll
                for (int u = 0; u < 4; u++) j[u] = bA.j[u];
            } else {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int t = t0 + 64 * u + l;
                    j[u] = t < ln ? ball_idx[st + t] : -1;
                }
            }
            int sj[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                c[u] = j[u] >= 0 ? claim[j[u]] : -1;
                // the label gather rides along with the claim gather (one round trip instead of two in a row; the chain
                // index -> claim -> label -> coherent re-read -> atomic bounds this kernel, not the request rate)
                sj[u] = (thr.mode == 0 && j[u] >= 0) ? (int)sem[j[u]] : lab;
                if (GROUPS && j[u] >= 0 && root[j[u]] != r) sj[u] = lab + 1;   // other group: never label-compatible
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (t0 + 64 * u >= ln) break;
                // stale reads are only ever too large -> a redundant atomic
                bool post = c[u] > p && sj[u] == lab;
                // the L2 of an XCD does not see the atomics of the other seven: a cached line keeps saying "unclaimed"
                // long after the node was taken, and every such edge would post an atomic (memory-side, expensive);
                // a coherent re-read filters them
                if (post) post = ld_agent(&claim[j[u]]) > p;
                if (post) atomicMin(&claim[j[u]], p);
                const unsigned long long m = __ballot(post);
                if (l == 0) am[(t0 >> 6) + u] = m;
            }
        }
        hA = hB; bA = bB; hB = hC; idC = idD;
    }
}

__global__ __launch_bounds__(256) void dir_win_kernel(int level, const int *__restrict__ ball_idx,
                                                       const int *__restrict__ start_len,
                                                       const int *__restrict__ root, const int *__restrict__ F,
                                                       int *counters, int *claim,
                                                       const unsigned long long *__restrict__ amask,
                                                       unsigned long long *tile_status, int *F_next, int *seg_start_next,
                                                       int *seg_cnt_next, int strict, int *guard)
{
    __shared__ unsigned long long s_wm[DIR_TILE][MAX_SLICES];
    __shared__ int s_cnt[DIR_TILE];
    __shared__ int s_off[DIR_TILE];
    __shared__ int s_bcast[2];
    __shared__ int s_node[DIR_TILE], s_st[DIR_TILE], s_ln[DIR_TILE];
    const int nF = counters[6 + (level & 1)];
    const int ntiles = (nF + DIR_TILE - 1) / DIR_TILE;
    const int l = lane_id(), wv = wave_id();
    constexpr int PER_WAVE = DIR_TILE / 4;
    constexpr unsigned long long FLAG_AGG = 1ull << 62, FLAG_INCL = 2ull << 62, VAL = (1ull << 62) - 1;
    if ((int)blockIdx.x >= ntiles) return;  // surplus workgroups leave without touching the ticket counter
    for (;;) {
        if (threadIdx.x == 0) s_bcast[0] = atomicAdd(&counters[8], 1);
        __syncthreads();
        const int tile = s_bcast[0];
        if (tile >= ntiles) break;
        if (threadIdx.x < DIR_TILE) {   // the tile's list headers, all positions at once
            const int pos = tile * DIR_TILE + threadIdx.x;
            int node = 0, st = 0, ln = 0;
            if (pos < nF) {
                node = F[pos];
                st = start_len[node * 2];
                ln = min(start_len[node * 2 + 1], 64 * MAX_SLICES);
            }
            s_node[threadIdx.x] = node; s_st[threadIdx.x] = st; s_ln[threadIdx.x] = ln;
        }
        __syncthreads();
        // ---- pass 1: winners per position, masks kept in LDS
        for (int k = 0; k < PER_WAVE; k++) {
            const int q = wv * PER_WAVE + k, pos = tile * DIR_TILE + q;
            int count = 0;
            if (pos < nF) {
                const int node = s_node[q];
                const int st = s_st[q], ln = s_ln[q];
                const unsigned long long *am = amask + mask_slot(st, node, strict);
                const int nsl = (ln + 63) >> 6;
                const unsigned long long mine = l < nsl ? am[l] : 0ull;   // all slice masks of the list in one load
                for (int c0 = 0; c0 < nsl; c0 += 4) {   // four slices in flight: index gathers, then claim gathers
                    int j[4], cl[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const unsigned long long m = __shfl(mine, (c0 + u) & 63, 64);
                        j[u] = (c0 + u < nsl && ((m >> l) & 1ull)) ? ball_idx[st + 64 * (c0 + u) + l] : -1;
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) cl[u] = j[u] >= 0 ? claim[j[u]] : -2;
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        if (c0 + u >= nsl) break;
                        const unsigned long long wm = __ballot(cl[u] == pos);
                        if (l == 0) s_wm[q][c0 + u] = wm;
                        count += __popcll(wm);
                    }
                }
            }
            if (l == 0) s_cnt[q] = count;
        }
        __syncthreads();
        // ---- tile scan + decoupled look-back over the preceding tiles
        if (wv == 0) {
            const int v = l < DIR_TILE ? s_cnt[l] : 0;
            const int incl = wave_incl_scan(v);
            if (l < DIR_TILE) s_off[l] = incl - v;
            const int total = __shfl(incl, 63, 64);
            // relaxed on purpose: the word carries its own payload, nothing else is ordered by it (an acquire / release
            // pair at agent scope invalidates / writes back the L2 on every poll: a 10x slower kernel)
            if (l == 0)
                __hip_atomic_store(&tile_status[tile], (tile == 0 ? FLAG_INCL : FLAG_AGG) | (unsigned long long)total,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // look-back, 64 predecessors per step: sum the aggregates down to the nearest tile whose inclusive prefix
            // is known (tile -1 counts as "inclusive 0")
            int base = 0;
            for (int hi = tile - 1; hi >= 0;) {
                const int t = hi - l;
                const unsigned long long sv = t >= 0 ? __hip_atomic_load(&tile_status[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                     : FLAG_INCL;
                const unsigned long long incl_m = __ballot((sv >> 62) == 2ull), inval_m = __ballot((sv >> 62) == 0ull);
                const int first = incl_m ? __ffsll((long long)incl_m) - 1 : 64;
                const unsigned long long need = first >= 63 ? ~0ull : ((2ull << first) - 1ull);
                if (inval_m & need) continue;  // a predecessor in the window has not published yet: poll again
                base += wave_sum(l <= first ? (int)(sv & VAL) : 0);
                if (first < 64) break;
                hi -= 64;
            }
            if (l == 0) {
                if (tile > 0)
                    __hip_atomic_store(&tile_status[tile], FLAG_INCL | (unsigned long long)(base + total), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                s_bcast[1] = base;
                if (tile == ntiles - 1) counters[6 + ((level + 1) & 1)] = base + total;
            }
        }
        __syncthreads();
        // ---- pass 2: winners out, in (position, slot) order
        const int base = s_bcast[1];
        for (int k = 0; k < PER_WAVE; k++) {
            const int q = wv * PER_WAVE + k, pos = tile * DIR_TILE + q;
            const int total = s_cnt[q];
            if (pos >= nF || total == 0) continue;
            const int node = s_node[q];
            const int st = s_st[q], ln = s_ln[q];
            const int nsl = (ln + 63) >> 6;
            int out = base + s_off[q];
            if (l == 0) {
                const int r = root[node];
                atomicMin(&seg_start_next[r], out);
                atomicAdd(&seg_cnt_next[r], total);
            }
            for (int c = 0; c < nsl; c++) {
                const unsigned long long wm = s_wm[q][c];
                if (wm == 0ull) continue;
                if ((wm >> l) & 1ull) {
                    const int j = ball_idx[st + 64 * c + l];
                    F_next[out + __popcll(wm & ((1ull << l) - 1ull))] = j;
                    // strict (see bfs_expand_kernel): a neighbour named twice in one list would be emitted twice
                    if (guard && atomicExch(&guard[j], 1) == 1) counters[17] = 1;
                    claim[j] = -1;  // visited
                }
                out += __popcll(wm);
            }
        }
        __syncthreads();
    }
}

// cluster of a seed = what the expansion reached; counters[9] counts the seeds whose group holds more than that
// (stage 1: the component has leftovers; stage 2: impossible, checked)
__global__ void dir_finish_kernel(const int *__restrict__ worklist, int *counters, const int *__restrict__ comp_size,
                                  const int *__restrict__ comp_base, const int *__restrict__ done,
                                  const int *__restrict__ seg_cnt, int *cl_size, int *cl_start, int *defi)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= counters[0]) return;
    const int r = worklist[w];
    const int reached = done[r] + seg_cnt[r];
    cl_size[r] = reached;
    cl_start[r] = comp_base[r];
    const int short_of = reached != comp_size[r];
    defi[r] = short_of;  // (not final while the frontier is alive: the host looks at counters[9] only once it is empty)
    if (short_of) atomicAdd(&counters[9], 1);
}

// ---- stage 2
// leftovers = unvisited points of the components stage 1 did not exhaust: own label, queued in left[] (counters[10])
__global__ void dir2_collect_kernel(int N, const int *__restrict__ root, const int *__restrict__ defi,
                                    const int *__restrict__ claim, int *lab, int *comp_size, int *left, int *counters)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const bool is_left = defi[root[i]] && claim[i] == INT_BIG;
    lab[i] = is_left ? i : -1;
    if (is_left) comp_size[i] = 0;
    const unsigned long long m = __ballot(is_left);
    if (m) {
        int base = 0;
        const int leader = __ffsll((long long)m) - 1;
        if (lane_id() == leader) base = atomicAdd(&counters[10], __popcll(m));
        base = __shfl(base, leader, 64);
        if (is_left) left[base + __popcll(m & ((1ull << lane_id()) - 1ull))] = i;
    }
}

// one round of min-propagation: every leftover pushes its label along its out-edges to leftovers with a larger one;
// counters[11 + (round & 1)] says whether anything moved (the other one is cleared for the next round)
__global__ __launch_bounds__(256) void dir2_propagate_kernel(Thr thr, int round, const int16_t *__restrict__ sem,
                                                              const int *__restrict__ ball_idx,
                                                              const int *__restrict__ start_len,
                                                              const int *__restrict__ left, int *lab, int *counters)
{
    const int nleft = counters[10];
    if (blockIdx.x == 0 && threadIdx.x == 0) counters[11 + ((round + 1) & 1)] = 0;
    const int waves = blockDim.x >> 6, l = lane_id();
    bool moved = false;
    for (int q = blockIdx.x * waves + wave_id(); q < nleft; q += gridDim.x * waves) {
        const int v = left[q];
        const int lv = ld_agent(&lab[v]);
        const int st = start_len[v * 2], ln = start_len[v * 2 + 1];
        const int sv = thr.mode == 0 ? (int)sem[v] : 0;
        for (int t = l; t < ln; t += 64) {
            const int u = ball_idx[st + t];
            if (thr.mode == 0 && (int)sem[u] != sv) continue;
            // a stale (cached) label is only ever too large: a redundant atomic at worst; visited points carry -1
            if (lab[u] > lv && atomicMin(&lab[u], lv) > lv) moved = true;
        }
    }
    if (__ballot(moved) && l == 0) counters[11 + (round & 1)] = 1;
}

// the labels become the groups: root[] rewritten for the leftovers, group sizes counted at the label
__global__ void dir2_count_kernel(const int *__restrict__ left, const int *lab, int *root, int *comp_size,
                                  int *counters)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q == 0) counters[0] = 0;  // the work list is rebuilt by the select kernel that follows
    if (q >= counters[10]) return;
    const int v = left[q];
    const int a = ld_agent(&lab[v]);
    root[v] = a;
    atomicAdd(&comp_size[a], 1);
}

__global__ void dir2_select_kernel(Thr thr, const int *__restrict__ left, const int *__restrict__ root,
                                   const int *__restrict__ comp_size, int *worklist, int *counters)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= counters[10]) return;
    const int v = left[q];
    if (root[v] == v && qualifies(thr, comp_size[v], v)) worklist[atomicAdd(&counters[0], 1)] = v;
}

// one cluster per qualifying component: seed = root, size = component size, members at comp_base
__global__ void glob_finish_kernel(const int *__restrict__ worklist, const int *__restrict__ counters,
                                   const int *__restrict__ comp_size, const int *__restrict__ comp_base, int *cl_size,
                                   int *cl_start)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= counters[0]) return;
    const int r = worklist[w];
    cl_size[r] = comp_size[r];
    cl_start[r] = comp_base[r];
}

__global__ void bfs_keep_kernel(int N, Thr thr, const int *__restrict__ cl_size, int *keep, int *keep_size)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int s = cl_size[i];
    const int k = (s > 0 && qualifies(thr, s, i)) ? 1 : 0;
    keep[i] = k;
    keep_size[i] = k ? s : 0;
}

// counters[3] = nCluster, counters[4] = sumNPoint (written by the scans)
__global__ void bfs_emit_kernel(int N, const int *__restrict__ counters, const int *__restrict__ scratch_node,
                                const int *__restrict__ scratch_seed, const int *__restrict__ cl_size,
                                const int *__restrict__ cl_start, const int *__restrict__ cid,
                                const int *__restrict__ out_off, const int *__restrict__ keep_size,
                                int *__restrict__ cluster_idxs, int *__restrict__ cluster_offsets)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int used = counters[2];
    if (t < used) {
        const int seed = scratch_seed[t];
        if (seed >= 0 && keep_size[seed] > 0) {
            const int pos = out_off[seed] + (t - cl_start[seed]);
            cluster_idxs[pos * 2 + 0] = cid[seed];
            cluster_idxs[pos * 2 + 1] = scratch_node[t];
        }
    }
    if (t < N && keep_size[t] > 0) cluster_offsets[cid[t]] = out_off[t];
    if (t == 0) cluster_offsets[counters[3]] = counters[4];
}

// 64-bit words of a node bitmap, padded to an even count (16-byte copies)
inline int bm_words64(int N) { return ((N + 63) / 64 + 1) & ~1; }

struct BfsWorkspace {
    int *parent, *root, *comp_size, *visited, *claim, *worklist, *scratch_node, *scratch_seed, *cl_size, *cl_start, *keep,
        *keep_size, *cid, *out_off, *counters, *Fa, *Fb, *comp_base, *done[2], *seg_start[2], *seg_cnt[2], *defi, *left, *guard;
    unsigned long long *vp, *cc, *tile_status, *amask, *bits;
    int *cand, *cnt;
    void *scan_ws;
};
size_t carve(BfsWorkspace &w, int N, void *base)
{
    size_t off = 0;
    auto take = [&](size_t bytes) {
        void *r = base ? (void *)((char *)base + off) : nullptr;
        off += ms3d_align(bytes);
        return (int *)r;
    };
    const size_t nb = sizeof(int) * (size_t)N;
    w.parent = take(nb); w.root = take(nb); w.comp_size = take(nb); w.visited = take(nb); w.claim = take(nb); w.worklist = take(nb);
    w.scratch_node = take(2 * nb); w.scratch_seed = take(2 * nb); w.cl_size = take(nb); w.cl_start = take(nb);
    w.defi = take(nb); w.left = take(nb); w.guard = take(nb);
    w.keep = take(nb); w.keep_size = take(nb); w.cid = take(nb); w.out_off = take(nb);
    // chip-wide expansion: frontier double buffer, per-component bookkeeping double-buffered by level parity, one
    // node words / candidate tags / claims / child counts, scan state per tile of 64 positions
    w.Fa = take(nb); w.Fb = take(nb);
    w.comp_base = take(nb);
    for (int k = 0; k < 2; k++) { w.done[k] = take(nb); w.seg_start[k] = take(nb); w.seg_cnt[k] = take(nb); }
    w.vp = (unsigned long long *)take(sizeof(unsigned long long) * (size_t)N);
    w.cc = (unsigned long long *)take(sizeof(unsigned long long) * (size_t)N);
    w.cand = take(nb); w.cnt = take(nb);
    w.tile_status = (unsigned long long *)take(sizeof(unsigned long long) * ((size_t)N / DIR_TILE + 2));
    w.amask = (unsigned long long *)take(sizeof(unsigned long long) * (size_t)N * (MAX_SLICES + 1));
    w.bits = (unsigned long long *)take(3 * sizeof(unsigned long long) * (size_t)bm_words64(N));   // bitmaps V, F, C
    w.counters = take(sizeof(int) * 64);   // [0..15] as documented at the kernels, [17] repeated member, [18] list > 1024, [32] validation
    w.scan_ws = take(ms3d_scan_workspace_bytes());
    return off;
}

}  // namespace

// also used by hais.hip (mode 0, threshold 0 = every connected component)
int ms3d_bfs_run_internal(int mode, int thr_i, float thr_f, int capped_hint, const int16_t *sem, const int *ball_idx,
                          long n_edges, const int *start_len, int N, int *cluster_idxs, int *cluster_offsets, int *counts,
                          void *workspace, size_t workspace_bytes, hipStream_t stream);

namespace {
std::atomic<int> g_dense_depth_hint{10};   // levels the last dense-graph expansion needed

int bfs_run(Thr thr, int capped_hint, const int16_t *sem, const int *ball_idx, long n_edges, const int *start_len, int N,
            int *cluster_idxs,
            int *cluster_offsets, int *counts, void *workspace, size_t workspace_bytes, hipStream_t stream)
{
    counts[0] = counts[1] = 0;
    if (N <= 0) {
        MS3D_CHECK(hipMemsetAsync(cluster_offsets, 0, sizeof(int), stream));
        return 0;
    }
    BfsWorkspace w;
    if (carve(w, N, workspace) > workspace_bytes) return MS3D_E_WORKSPACE;
    const int nb = ms3d_divup(N, 256);
    // capped_hint < 0: nobody vouches for this graph (round 5; the reference's host BFS accepts ANY adjacency lists,
    // bfs_cluster.cpp:28-54).  Then nothing is assumed about it: the lists are validated first (headers inside the edge
    // array, targets inside [0, N): MS3D_E_UNSUPPORTED otherwise -- the reference would read out of bounds), no symmetry
    // is assumed (a dense graph takes the chip-wide DIRECTED expansion, which is out-edge reachability from ascending seeds
    // for any graph; lists beyond 1024 entries take the per-component replay), and a neighbour named twice in one list
    // (which the serial loop skips and the parallel claims would emit twice) fails the call.
    const bool strict = capped_hint < 0;
    if (strict) {
        int bad = 0;
        MS3D_CHECK(hipMemsetAsync(w.counters + 32, 0, sizeof(int), stream));
        const long work = n_edges > N ? n_edges : N;
        bfs_validate_kernel<<<(int)std::min<long>((work + 255) / 256, 256 * 16), 256, 0, stream>>>(N, n_edges, ball_idx, start_len,
                                                                                                  w.counters + 32);
        MS3D_LAUNCH_CHECK();
        MS3D_CHECK(hipMemcpyAsync(&bad, w.counters + 32, sizeof(int), hipMemcpyDeviceToHost, stream));
        MS3D_CHECK(hipStreamSynchronize(stream));
        if (bad) { if (getenv("MS3D_DEBUG")) fprintf(stderr, "[bfs] validation failed\n"); return MS3D_E_UNSUPPORTED; }
        MS3D_CHECK(hipMemsetAsync(w.guard, 0, sizeof(int) * (size_t)N, stream));
    }
    int *guard = strict ? w.guard : nullptr;
    // weak components, their sizes, and the work list of the components that can hold a qualifying cluster
    auto prepare = [&]() -> int {
        bfs_init_kernel<<<nb, 256, 0, stream>>>(N, thr, sem, ball_idx, start_len, w.parent, w.comp_size, w.visited, w.claim,
                                               w.cl_size, w.scratch_seed, w.defi, w.counters);
        MS3D_LAUNCH_CHECK();
        bfs_compress_kernel<<<nb, 256, 0, stream>>>(N, w.parent);
        MS3D_LAUNCH_CHECK();
        bfs_link_kernel<<<nb, 256, 0, stream>>>(N, thr, sem, ball_idx, start_len, w.parent);
        MS3D_LAUNCH_CHECK();
        bfs_compress_kernel<<<nb, 256, 0, stream>>>(N, w.parent);
        MS3D_LAUNCH_CHECK();
        bfs_hook_kernel<<<min(ms3d_divup(N, 4), 256 * 32), 256, 0, stream>>>(N, thr, capped_hint == 0 ? 1 : 0, sem, ball_idx,
                                                                             start_len, w.parent);
        MS3D_LAUNCH_CHECK();
        bfs_compress_kernel<<<nb, 256, 0, stream>>>(N, w.parent);
        MS3D_LAUNCH_CHECK();
        bfs_flatten_kernel<<<nb, 256, 0, stream>>>(N, w.parent, w.root, w.comp_size, start_len, w.counters);
        MS3D_LAUNCH_CHECK();
        bfs_select_kernel<<<nb, 256, 0, stream>>>(N, thr, w.root, w.comp_size, w.worklist, w.counters);
        MS3D_LAUNCH_CHECK();
        return 0;
    };
    {
        const int rc0 = prepare();
        if (rc0) return rc0;
    }
    // Dense symmetric graphs (shifted coordinates: hundreds of neighbours per point, a handful of BFS levels)
    // and dense directed graphs (capped lists) are expanded by the whole chip level by level; sparse graphs by the
    // per-component replay kernel.
    // When the ball query already told us that no list was capped (capped_hint == 0) nothing has to be read back
    // before the expansion; the frontier size is checked together with the final counts.
    static const bool dbg = getenv("MS3D_DEBUG") != nullptr;
#define DBG(tag)                                                                                             \
    if (dbg) {                                                                                               \
        hipError_t e_ = hipStreamSynchronize(stream);                                                        \
        fprintf(stderr, "[bfs] %s N=%d edges=%ld mode=%d hint=%d err=%d\n", tag, N, n_edges, thr.mode, capped_hint, (int)e_); \
    }
    DBG("after select");
    bool replay = true, dense = false, directed = false, stage2 = false;
    // LDS-resident node bitmaps in the dense symmetric passes: up to ~1.1 M points (144 KB of bits per workgroup)
    static const bool bm_on = [] { const char *e = getenv("MS3D_BFS_BITMAP"); return !e || atoi(e) != 0; }();
    static const hipError_t bm_attr = [] {
        hipError_t e = hipFuncSetAttribute((const void *)glob_mark_bm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)glob_pull_bm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)glob_win_bm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
        return e;
    }();
    const bool bitmaps = bm_on && bm_attr == hipSuccess && (size_t)bm_words64(N) * 8 <= 144 * 1024;
    const int wl_grid = ms3d_divup(N, 256);  // per-work-item kernels are launched for the upper bound N, they mask on nwork
    int level = 0;
    auto run_levels = [&](int nlev) -> int {
        for (int it = 0; it < nlev; it++, level++) {
            const int c = level & 1, n = c ^ 1;
            int *Fc = c ? w.Fb : w.Fa, *Fn = c ? w.Fa : w.Fb;
            if (directed) {
                auto claim_fn = stage2 ? dir_claim_kernel<true> : dir_claim_kernel<false>;
                claim_fn<<<256 * 8, 256, 0, stream>>>(thr, level, sem, ball_idx, start_len, w.root, Fc, w.counters,
                                                             w.worklist, w.comp_base, w.done[c], w.done[n], w.seg_start[c],
                                                             w.seg_start[n], w.seg_cnt[c], w.seg_cnt[n], w.claim,
                                                             w.scratch_node, w.scratch_seed, w.amask, w.tile_status,
                                                             strict ? 1 : 0);
                MS3D_LAUNCH_CHECK();
                dir_win_kernel<<<256 * 8, 256, 0, stream>>>(level, ball_idx, start_len, w.root, Fc, w.counters, w.claim, w.amask,
                                                           w.tile_status, Fn, w.seg_start[n], w.seg_cnt[n], strict ? 1 : 0, guard);
                MS3D_LAUNCH_CHECK();
                continue;
            }
            if (bitmaps) {
                const int nw = bm_words64(N);
                unsigned long long *Vb = w.bits, *Fb = w.bits + nw, *Cb = w.bits + 2 * (size_t)nw;
                const size_t lds = (size_t)nw * 8;
                glob_bits_kernel<<<nb, 256, 0, stream>>>(N, level, w.vp, Vb, Fb, Cb);
                MS3D_LAUNCH_CHECK();
                glob_mark_bm_kernel<<<256 * 2, BM_THREADS, lds, stream>>>(level, ball_idx, start_len, w.root, Fc, w.counters,
                                                                        w.worklist, w.comp_base, w.done[c], w.done[n],
                                                                        w.seg_start[c], w.seg_start[n], w.seg_cnt[c],
                                                                        w.seg_cnt[n], w.cand, w.cnt, w.scratch_node,
                                                                        w.scratch_seed, w.tile_status, Vb, nw);
                MS3D_LAUNCH_CHECK();
                glob_pull_bm_kernel<<<256 * 2, BM_THREADS, lds, stream>>>(N, level, ball_idx, start_len, w.vp, w.cand, w.cc, w.cnt,
                                                                        w.counters, Fb, reinterpret_cast<unsigned *>(Cb), nw);
                MS3D_LAUNCH_CHECK();
                glob_win_bm_kernel<<<256 * 2, WIN_BM_THREADS, lds, stream>>>(level, ball_idx, start_len, w.root, Fc, w.counters, w.cnt,
                                                                       w.cc, w.vp, w.tile_status, Fn, w.seg_start[n],
                                                                       w.seg_cnt[n], Cb, nw);
                MS3D_LAUNCH_CHECK();
                continue;
            }
            glob_mark_kernel<<<256 * 8, 256, 0, stream>>>(level, ball_idx, start_len, w.root, Fc, w.counters, w.worklist,
                                                         w.comp_base, w.done[c], w.done[n], w.seg_start[c], w.seg_start[n],
                                                         w.seg_cnt[c], w.seg_cnt[n], w.vp, w.cand, w.cnt, w.scratch_node,
                                                         w.scratch_seed, w.tile_status);
            MS3D_LAUNCH_CHECK();
            glob_pull_kernel<<<nb, 256, 0, stream>>>(N, level, ball_idx, start_len, w.vp, w.cand, w.cc, w.cnt);
            MS3D_LAUNCH_CHECK();
            glob_win_kernel<<<256 * 8, 256, 0, stream>>>(level, ball_idx, start_len, w.root, Fc, w.counters, w.cnt, w.cc, w.vp,
                                                        w.tile_status, Fn, w.seg_start[n], w.seg_cnt[n]);
            MS3D_LAUNCH_CHECK();
        }
        return 0;
    };
    auto launch_replay = [&]() -> int {
        static const int force_nt = [] { const char *e = getenv("MS3D_BFS_REPLAY_NT"); return e ? atoi(e) : 0; }();
        if (force_nt == 1024 || (force_nt == 0 && n_edges >= (long)N * 24))
            bfs_expand_kernel<1024><<<256 * 2, 1024, 0, stream>>>(N, thr, sem, ball_idx, start_len, w.root, w.comp_size,
                                                              w.worklist, w.counters, w.visited, w.claim, w.scratch_node,
                                                              w.scratch_seed, w.cl_size, w.cl_start, guard);
        else
            bfs_expand_kernel<512><<<256 * 2, 512, 0, stream>>>(N, thr, sem, ball_idx, start_len, w.root, w.comp_size,
                                                              w.worklist, w.counters, w.visited, w.claim, w.scratch_node,
                                                              w.scratch_seed, w.cl_size, w.cl_start, guard);
        MS3D_LAUNCH_CHECK();
        return 0;
    };
    static const bool dir_on = [] { const char *e = getenv("MS3D_BFS_DIRECTED"); return !e || atoi(e) != 0; }();
    if (n_edges >= (long)N * 24) {
        int capped = capped_hint;
        bool too_long = false;
        if (strict) {
            int h[32];
            MS3D_CHECK(hipMemcpyAsync(h, w.counters, sizeof(int) * 32, hipMemcpyDeviceToHost, stream));
            MS3D_CHECK(hipStreamSynchronize(stream));
            capped = 1;               // unknown symmetry: the directed expansion is right for any graph
            too_long = h[18] != 0;    // ... unless a list outgrows its edge masks: the replay
        }
        if (too_long) {
            // (replay stays true)
        } else if (capped == 0) {
            replay = false;
            dense = true;
            glob_vp_init_kernel<<<nb, 256, 0, stream>>>(N, thr, sem, w.vp, w.cand, w.cc);
            MS3D_LAUNCH_CHECK();
            glob_init_kernel<<<wl_grid, 256, 0, stream>>>(w.worklist, w.comp_size, w.counters, w.Fa, w.comp_base, w.done[0],
                                                         w.seg_start[0], w.seg_cnt[0], w.vp);
            MS3D_LAUNCH_CHECK();
            DBG("after glob_init");
        } else if (dir_on) {
            replay = false;
            directed = true;
            dir_init_kernel<<<wl_grid, 256, 0, stream>>>(w.worklist, w.comp_size, w.counters, w.Fa, w.comp_base, w.done[0],
                                                        w.seg_start[0], w.seg_cnt[0], w.claim);
            MS3D_LAUNCH_CHECK();
        }
    }
    if (replay) {
        const int rc2 = launch_replay();
        if (rc2) return rc2;
    }
    int host[32];
    for (;;) {
        if (dense) {
            // the levels are launched speculatively (a shifted-coordinate blob is exhausted after ~10) and the frontier
            // counter is read together with the final counts; a deeper component simply gets 8 more levels -- the
            // assembly below is idempotent
            if (level > 60000) return MS3D_E_UNSUPPORTED;  // 16-bit level field of the node word
            // how many levels to launch blind: what the last dense graph needed + 2 (an exhausted level still costs three
            // ~5 us launches; 16 fixed levels spent ~0.2 ms per PointGroup step on empty ones)
            const int spec = level == 0 ? min(24, max(6, g_dense_depth_hint.load() + 2)) : 8;
            int rc2 = run_levels(spec);
            if (rc2) return rc2;
            DBG("after levels");
            glob_finish_kernel<<<wl_grid, 256, 0, stream>>>(w.worklist, w.counters, w.comp_size, w.comp_base, w.cl_size,
                                                           w.cl_start);
            MS3D_LAUNCH_CHECK();
        }
        if (directed) {
            int rc2 = run_levels(8);   // a capped list spans its blob: 2-4 levels
            if (rc2) return rc2;
            MS3D_CHECK(hipMemsetAsync(w.counters + 9, 0, sizeof(int), stream));
            dir_finish_kernel<<<wl_grid, 256, 0, stream>>>(w.worklist, w.counters, w.comp_size, w.comp_base, w.done[level & 1],
                                                          w.seg_cnt[level & 1], w.cl_size, w.cl_start, w.defi);
            MS3D_LAUNCH_CHECK();
        }
        bfs_keep_kernel<<<nb, 256, 0, stream>>>(N, thr, w.cl_size, w.keep, w.keep_size);
        MS3D_LAUNCH_CHECK();
        int rc = ms3d_exclusive_scan_i32(w.keep, w.cid, N, w.counters + 3, w.scan_ws, stream);
        if (rc) return rc;
        rc = ms3d_exclusive_scan_i32(w.keep_size, w.out_off, N, w.counters + 4, w.scan_ws, stream);
        if (rc) return rc;
        bfs_emit_kernel<<<2 * nb, 256, 0, stream>>>(N, w.counters, w.scratch_node, w.scratch_seed, w.cl_size, w.cl_start, w.cid,
                                               w.out_off, w.keep_size, cluster_idxs, cluster_offsets);
        MS3D_LAUNCH_CHECK();
        DBG("after emit");
        MS3D_CHECK(hipMemcpyAsync(host, w.counters, sizeof(int) * 32, hipMemcpyDeviceToHost, stream));
        MS3D_CHECK(hipStreamSynchronize(stream));
        if (strict && host[17]) {                                 // a neighbour named twice in one list
            if (dbg) fprintf(stderr, "[bfs] repeated member flagged\n");
            return MS3D_E_UNSUPPORTED;
        }
        if (dbg) fprintf(stderr, "[bfs] counters %d %d %d %d %d %d %d %d | %d %d level=%d\n", host[0], host[1], host[2], host[3], host[4], host[5], host[6], host[7], host[8], host[9], level);
        if (dense) {
            if (host[6 + (level & 1)] == 0) {       // frontier empty: every component was exhausted
                g_dense_depth_hint.store(host[14]);
                break;
            }
            continue;
        }
        if (directed) {
            if (host[6 + (level & 1)] != 0) continue;     // deeper than the levels launched so far
            if (host[9] == 0) break;                      // every group was exhausted by its seed: done
            if (stage2) return MS3D_E_INTERNAL;           // a label group its own seed does not reach: cannot happen
            // stage 2: the leftovers of the components that hold more than the cluster of their root
            stage2 = true;
            int *lab = w.visited;
            dir2_collect_kernel<<<nb, 256, 0, stream>>>(N, w.root, w.defi, w.claim, lab, w.comp_size, w.left, w.counters);
            MS3D_LAUNCH_CHECK();
            for (int round = 0;;) {
                for (int k = 0; k < 4; k++, round++) {
                    dir2_propagate_kernel<<<256 * 4, 256, 0, stream>>>(thr, round, sem, ball_idx, start_len, w.left, lab,
                                                                      w.counters);
                    MS3D_LAUNCH_CHECK();
                }
                MS3D_CHECK(hipMemcpyAsync(host, w.counters, sizeof(int) * 32, hipMemcpyDeviceToHost, stream));
                MS3D_CHECK(hipStreamSynchronize(stream));
                if (host[11 + ((round - 1) & 1)] == 0) break;   // the last round moved nothing: fixed point
            }
            const int left_grid = ms3d_divup(host[10] > 0 ? host[10] : 1, 256);
            dir2_count_kernel<<<left_grid, 256, 0, stream>>>(w.left, lab, w.root, w.comp_size, w.counters);
            MS3D_LAUNCH_CHECK();
            dir2_select_kernel<<<left_grid, 256, 0, stream>>>(thr, w.left, w.root, w.comp_size, w.worklist, w.counters);
            MS3D_LAUNCH_CHECK();
            level = 0;
            dir_init_kernel<<<left_grid, 256, 0, stream>>>(w.worklist, w.comp_size, w.counters, w.Fa, w.comp_base, w.done[0],
                                                          w.seg_start[0], w.seg_cnt[0], w.claim);
            MS3D_LAUNCH_CHECK();
            continue;
        }
        break;
    }
    counts[0] = host[3];
    counts[1] = host[4];
    return 0;
}

}  // namespace

int ms3d_bfs_run_internal(int mode, int thr_i, float thr_f, int capped_hint, const int16_t *sem, const int *ball_idx,
                          long n_edges, const int *start_len, int N, int *cluster_idxs, int *cluster_offsets, int *counts,
                          void *workspace, size_t workspace_bytes, hipStream_t stream)
{
    Thr thr{mode, thr_i, thr_f, nullptr, nullptr};
    return bfs_run(thr, capped_hint, sem, ball_idx, n_edges, start_len, N, cluster_idxs, cluster_offsets, counts, workspace,
                   workspace_bytes, stream);
}

extern "C" {

size_t ms3d_bfs_workspace_bytes(int N)
{
    BfsWorkspace w;
    return carve(w, N > 0 ? N : 1, nullptr);
}

int ms3d_pg_bfs_cluster(const int16_t *semantic_label, const int *ball_query_idxs, long n_edges, const int *start_len,
                        int N, int threshold, int capped_hint, int *cluster_idxs, int *cluster_offsets, int *counts, void *workspace,
                        size_t workspace_bytes, ms3d_stream_t stream)
{
    Thr thr{0, threshold, 0.f, nullptr, nullptr};
    return bfs_run(thr, capped_hint, semantic_label, ball_query_idxs, n_edges, start_len, N, cluster_idxs, cluster_offsets, counts,
                   workspace, workspace_bytes, (hipStream_t)stream);
}

int ms3d_sg_bfs_cluster(const float *class_numpoint_mean, const int *ball_query_idxs, long n_edges,
                        const int *start_len, int N, float threshold, int capped_hint, int class_id, int *cluster_idxs, int *cluster_offsets, int *counts,
                        void *workspace, size_t workspace_bytes, ms3d_stream_t stream)
{
    const float m = class_numpoint_mean[class_id];  // bfs_cluster.cpp:113-120
    Thr thr{1, 0, (m == -1.f) ? threshold : threshold * m, nullptr, nullptr};
    return bfs_run(thr, capped_hint, nullptr, ball_query_idxs, n_edges, start_len, N, cluster_idxs, cluster_offsets, counts, workspace,
                   workspace_bytes, (hipStream_t)stream);
}


// SoftGroup's per-class loop (model/softgroup.py:43-76) as ONE launch sequence: the caller concatenates the points of
// all classes in class order, ball-queries them with group id = class*B + scene, and passes each group's threshold
// (threshold or threshold*class_numpoint_mean[class], bfs_cluster.cpp:113-120).  Clusters come out by ascending seed,
// i.e. class-major -- exactly the order in which the reference concatenates its per-class results.
int ms3d_sg_bfs_cluster_batched(const uint8_t *group_of_point, const float *thr_per_group, const int *ball_query_idxs,
                                long n_edges, const int *start_len, int N, int capped_hint, int *cluster_idxs,
                                int *cluster_offsets, int *counts, void *workspace, size_t workspace_bytes,
                                ms3d_stream_t stream)
{
    Thr thr{2, 0, 0.f, group_of_point, thr_per_group};
    return bfs_run(thr, capped_hint, nullptr, ball_query_idxs, n_edges, start_len, N, cluster_idxs, cluster_offsets, counts, workspace,
                   workspace_bytes, (hipStream_t)stream);
}

}  // extern "C"
