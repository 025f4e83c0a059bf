// collapse8.hip.h — BVH2 -> 8-wide compressed nodes on the device (included by sah_build.hip and lbvh.hip, each with its own view of its BVH2).
//
// The device form of build_bvh8_host (bvh_host.cpp) and the replacement of MBVH::construct (backends/gpu-rt/src/lib.rs:1581, :1411): the
// collapse that minimises the SAH cost of the wide tree (Ylitie, Karras, Laine 2017, section 3.1) in two kernels, no host round trip:
//   k_dp_up   bottom-up over the BVH2: C(n, i) = cheapest way to stand for subtree n with at most i slots of a wide parent, i = 1..7, and the
//             split that achieves it.  A thread starts at every leaf and climbs; the second child to arrive at a node computes the node
//             (agent-scope fences around the arrival counter, as the fit kernels of lbvh.hip do).
//   k_emit8   top-down: one work item per wide node (BVH2 root, wide index) in a queue that the kernel itself fills — a thread owns the
//             queue slots t, t + T, ... and POLLS its next slot once per loop iteration (never in an inner wait loop: lanes of one wavefront
//             produce work for each other); an item is one 8-byte store / load at agent scope, the queue's (allocated, finished) pair one
//             64-bit word, so a lane that finds its slot empty and allocated == finished knows that no item will ever arrive.
// The interior children of a wide node get consecutive node indices and its primitives consecutive places in the output order (bvh8.h).
#pragma once
#include <hip/hip_runtime.h>

#include "bvh8.h"

namespace rfwhip {
namespace collapse8 {

constexpr uint32_t kNoNode = 0xffffffffu;
constexpr unsigned long long kEmptyItem = ~0ull;
constexpr uint32_t kSpinLimit = 1u << 22; // polls of one empty slot before a lane gives up and reports (a bug or a lost device, never a tree)

struct Cell {          // 32 B per BVH2 node
    float c[7];        // C(n, 1..7)
    uint32_t code;     // bits 3(i-2) .. 3(i-2)+2: slots given to the left child when i = 2..8 slots are split (0: use one slot fewer); bit 31: C(n, 1) is a leaf
};
struct Counters {
    unsigned long long alloc_done; // (items allocated) << 32 | items finished
    uint32_t nodes8;               // wide nodes allocated (the root is node 0)
    uint32_t prims8;               // places of the output order handed out
    uint32_t error;                // a lane gave up polling
    uint32_t pad;
};

struct Workspace {
    Cell* cells;
    uint32_t* below;   // primitives below node n
    uint32_t* first;   // first of them in the builder's order
    uint32_t* arrive;
    unsigned long long* queue;
    Counters* ctr;
    uint32_t queue_cap;
};
inline size_t align256(size_t v) { return (v + 255) / 256 * 256; }
inline size_t workspace_bytes(uint32_t n_nodes2, uint32_t n_prims)
{
    const size_t m = n_nodes2 ? n_nodes2 : 1, q = n_prims ? n_prims : 1;
    return align256(m * sizeof(Cell)) + 3 * align256(m * 4) + align256(q * 8) + align256(sizeof(Counters));
}
inline Workspace carve(void* base, uint32_t n_nodes2, uint32_t n_prims)
{
    const size_t m = n_nodes2 ? n_nodes2 : 1, q = n_prims ? n_prims : 1;
    char* w = static_cast<char*>(base);
    Workspace ws;
    ws.cells = reinterpret_cast<Cell*>(w); w += align256(m * sizeof(Cell));
    ws.below = reinterpret_cast<uint32_t*>(w); w += align256(m * 4);
    ws.first = reinterpret_cast<uint32_t*>(w); w += align256(m * 4);
    ws.arrive = reinterpret_cast<uint32_t*>(w); w += align256(m * 4);
    ws.queue = reinterpret_cast<unsigned long long*>(w); w += align256(q * 8);
    ws.ctr = reinterpret_cast<Counters*>(w);
    ws.queue_cap = (uint32_t)q;
    return ws;
}

__device__ inline float half_area3(const float* lo, const float* hi)
{
    const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    if (!(ex >= 0.0f) || !(ey >= 0.0f) || !(ez >= 0.0f)) return 0.0f;
    return ex * ey + ey * ez + ez * ex;
}

// A View names the BVH2 of one builder.  Node ids are whatever the builder uses (not necessarily dense); it provides
//   uint32_t size()                      ids are < size()
//   bool     valid(id)                   the id is a node of the tree
//   bool     is_leaf(id);  uint32_t leaf_first(id), leaf_count(id)   (count <= kMaxLeaf8)
//   uint32_t left(id), right(id), parent(id)   (parent of the root: kNoNode);  uint32_t root()
//   void     box(id, float* lo, float* hi)

template <class View> __device__ inline void dp_node(const View& v, uint32_t n, const Cell* cells, const uint32_t* below, const uint32_t* first, float prim_cost,
                                                     Cell& out, uint32_t& out_below, uint32_t& out_first)
{
    float lo[3], hi[3];
    v.box(n, lo, hi);
    const float area = half_area3(lo, hi);
    if (v.is_leaf(n)) {
        const uint32_t cnt = v.leaf_count(n);
        for (int i = 0; i < 7; i++) out.c[i] = area * (float)cnt * prim_cost;
        out.code = 0x80000000u;
        out_below = cnt;
        out_first = v.leaf_first(n);
        return;
    }
    const uint32_t l = v.left(n), r = v.right(n);
    // the children's cells were written by other threads: read them past the (never refreshed) vector L1
    float cl[7], cr[7];
    for (int i = 0; i < 7; i++) {
        cl[i] = __hip_atomic_load(&cells[l].c[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cr[i] = __hip_atomic_load(&cells[r].c[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const uint32_t bl = __hip_atomic_load(&below[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), br = __hip_atomic_load(&below[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t fl = __hip_atomic_load(&first[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), fr = __hip_atomic_load(&first[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    out_below = bl + br;
    out_first = fl < fr ? fl : fr;
    uint32_t code = 0u;
    auto dist = [&](int j, uint32_t& m_out) { // best split of j slots over the two children (each child takes 1..7)
        float best = INFINITY;
        m_out = 1u;
        for (int m = 1; m < j; m++) {
            const int a = m > 7 ? 7 : m, b = (j - m) > 7 ? 7 : (j - m);
            const float c = cl[a - 1] + cr[b - 1];
            if (c < best) { best = c; m_out = (uint32_t)m; }
        }
        return best;
    };
    uint32_t m;
    const float internal = area + dist(8, m);
    code |= m << 18; // i = 8
    const float leaf = out_below <= (uint32_t)kMaxLeaf8 ? area * (float)out_below * prim_cost : INFINITY;
    if (leaf <= internal) code |= 0x80000000u;
    out.c[0] = leaf <= internal ? leaf : internal;
    for (int i = 2; i < 8; i++) {
        const float d = dist(i, m);
        if (d < out.c[i - 2]) { out.c[i - 1] = d; code |= m << (3 * (i - 2)); }
        else out.c[i - 1] = out.c[i - 2]; // split code 0: use one slot fewer
    }
    out.code = code;
}

template <class View>
__global__ __launch_bounds__(256) void k_dp_up(const View v, Cell* cells, uint32_t* below, uint32_t* first, uint32_t* arrive, const float prim_cost)
{
    uint32_t n = blockIdx.x * 256u + threadIdx.x;
    if (n >= v.size() || !v.valid(n) || !v.is_leaf(n)) return;
    for (;;) {
        Cell c;
        uint32_t b, f;
        dp_node(v, n, cells, below, first, prim_cost, c, b, f);
        for (int i = 0; i < 7; i++) __hip_atomic_store(&cells[n].c[i], c.c[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&cells[n].code, c.code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&below[n], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&first[n], f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t p = v.parent(n);
        if (p == kNoNode) return;
        // publish, then arrive: the second arrival at a node owns it and reads both children
        __threadfence();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t old = atomicAdd(&arrive[p], 1u);
        if (old == 0u) return;
        __threadfence();
        n = p;
    }
}

// the slots of the wide node rooted at BVH2 node n (an interior node), following the recorded splits
template <class View> __device__ inline int slots_of(const View& v, const Cell* cells, const uint32_t* below, const uint32_t* first, uint32_t n, Slot8* out)
{
    struct F { uint32_t node; int i; };
    F st[16];
    int sp = 0, ns = 0;
    const int m8 = (int)((cells[n].code >> 18) & 7u);
    st[sp++] = {v.right(n), 8 - m8};
    st[sp++] = {v.left(n), m8};
    while (sp > 0) {
        const F f = st[--sp];
        const uint32_t code = cells[f.node].code;
        const bool leaf2 = v.is_leaf(f.node);
        int i = f.i > 7 ? 7 : f.i;
        while (!leaf2 && i > 1 && ((code >> (3 * (i - 2))) & 7u) == 0u) i--;
        if (leaf2 || i == 1) {
            Slot8 s;
            v.box(f.node, s.lo, s.hi);
            s.ref = (code & 0x80000000u) ? make_leaf(first[f.node], below[f.node]) : f.node;
            if (ns < 8) out[ns++] = s;
            continue;
        }
        const int mm = (int)((code >> (3 * (i - 2))) & 7u);
        if (sp + 2 <= 16) {
            st[sp++] = {v.right(f.node), i - mm};
            st[sp++] = {v.left(f.node), mm};
        }
    }
    return ns;
}

template <class View>
__global__ __launch_bounds__(256) void k_emit8(const View v, const Cell* __restrict__ cells, const uint32_t* __restrict__ below, const uint32_t* __restrict__ first,
                                              const uint32_t* __restrict__ order_in, Node8* __restrict__ nodes_out, uint32_t* __restrict__ order_out, Counters* ctr,
                                              unsigned long long* queue, const uint32_t queue_cap)
{
    const uint32_t T = gridDim.x * 256u;
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    uint32_t spins = 0;
    while (i < queue_cap) {
        const unsigned long long item = __hip_atomic_load(&queue[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (item == kEmptyItem) {
            const unsigned long long ad = __hip_atomic_load(&ctr->alloc_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t alloc = (uint32_t)(ad >> 32), done = (uint32_t)ad;
            if (alloc == done && i >= alloc) break; // everything allocated is finished and this slot was never handed out: nothing will come
            if (++spins > kSpinLimit) { ctr->error = 1u; break; }
            __builtin_amdgcn_s_sleep(2);
            continue;
        }
        spins = 0;
        const uint32_t n2 = (uint32_t)(item >> 32), n8 = (uint32_t)item;
        Slot8 slots[8];
        const int ns = slots_of(v, cells, below, first, n2, slots);
        int pos_of[8];
        assign_octants(slots, ns, pos_of);
        const Slot8* by_pos[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        uint32_t ni = 0, nt = 0;
        for (int k = 0; k < ns; k++) {
            by_pos[pos_of[k]] = &slots[k];
            if (slots[k].ref & kLeafBit) nt += ((slots[k].ref >> 27) & 15u) + 1u;
            else ni++;
        }
        const uint32_t child_base = ni ? atomicAdd(&ctr->nodes8, ni) : 0u;
        const uint32_t tri_base = nt ? atomicAdd(&ctr->prims8, nt) : 0u;
        uint32_t leaf_offset[8];
        const Node8 nd = encode_node8(by_pos, child_base, tri_base, leaf_offset);
        uint4* dst = reinterpret_cast<uint4*>(nodes_out + n8);
        const uint4* src = reinterpret_cast<const uint4*>(&nd);
        for (int k = 0; k < 5; k++) dst[k] = src[k];
        uint32_t qbase = 0;
        if (ni) qbase = (uint32_t)(atomicAdd(&ctr->alloc_done, (unsigned long long)ni << 32) >> 32);
        uint32_t rank = 0;
        for (int p = 0; p < 8; p++) {
            if (!by_pos[p]) continue;
            const uint32_t ref = by_pos[p]->ref;
            if (ref & kLeafBit) {
                const uint32_t f = ref & kLeafFirstMask, cnt = ((ref >> 27) & 15u) + 1u;
                for (uint32_t k = 0; k < cnt; k++) order_out[tri_base + leaf_offset[p] + k] = order_in[f + k];
            } else {
                if (qbase + rank < queue_cap)
                    __hip_atomic_store(&queue[qbase + rank], ((unsigned long long)ref << 32) | (child_base + rank), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else ctr->error = 2u;
                rank++;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the items are out before the job counts as finished
        atomicAdd(&ctr->alloc_done, 1ull);
        i += T;
    }
}

// root handling + queue initialisation: a tree that is a single leaf (or empty) becomes its root node right here
template <class View>
__global__ void k_collapse_init(const View v, const uint32_t n_prims, const uint32_t* __restrict__ order_in, Node8* nodes_out, uint32_t* order_out, Counters* ctr,
                                unsigned long long* queue, uint32_t* node_count_out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    ctr->nodes8 = 1u;
    ctr->prims8 = 0u;
    ctr->error = 0u;
    const Slot8* by_pos[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    uint32_t off[8];
    Slot8 s;
    const uint32_t root = v.root();
    if (n_prims == 0u || v.is_leaf(root)) {
        uint32_t cnt = 0;
        if (n_prims != 0u) {
            v.box(root, s.lo, s.hi);
            cnt = v.leaf_count(root);
            s.ref = make_leaf(0u, cnt);
            by_pos[0] = &s;
            const uint32_t f = v.leaf_first(root);
            for (uint32_t k = 0; k < cnt; k++) order_out[k] = order_in[f + k];
        }
        nodes_out[0] = encode_node8(by_pos, 1u, 0u, off);
        ctr->prims8 = cnt;
        ctr->alloc_done = 0ull;
        return;
    }
    queue[0] = ((unsigned long long)root << 32) | 0u;
    ctr->alloc_done = 1ull << 32;
    (void)node_count_out;
}
template <class View> __global__ void k_collapse_finish(const Counters* ctr, uint32_t* node_count_out, uint32_t* error_out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (node_count_out) *node_count_out = ctr->nodes8;
    if (error_out && ctr->error) *error_out = ctr->error;
}

// BVH2 (through its view) -> nodes_out / order_out, all on `s`.  n_nodes2 = v.size(); order_in = the builder's primitive order.
template <class View>
inline hipError_t run(hipStream_t s, const View& v, uint32_t n_nodes2, uint32_t n_prims, const uint32_t* order_in, void* workspace, Node8* nodes_out,
                      uint32_t* order_out, uint32_t* node_count_out, uint32_t* error_out, float prim_cost)
{
    const Workspace ws = carve(workspace, n_nodes2, n_prims);
    hipError_t e = hipMemsetAsync(ws.arrive, 0, (size_t)(n_nodes2 ? n_nodes2 : 1) * 4, s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(ws.queue, 0xff, (size_t)ws.queue_cap * 8, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_collapse_init<View>), dim3(1), dim3(64), 0, s, v, n_prims, order_in, nodes_out, order_out, ws.ctr, ws.queue, node_count_out);
    if (n_prims > 0u) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_dp_up<View>), dim3((n_nodes2 + 255u) / 256u), dim3(256), 0, s, v, ws.cells, ws.below, ws.first, ws.arrive, prim_cost);
        // every thread of the grid has to be resident (a lane waits for items other lanes produce): at most one 256-thread workgroup per CU
        uint32_t groups = (ws.queue_cap / 4u + 255u) / 256u;
        groups = groups < 1u ? 1u : (groups > 256u ? 256u : groups);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_emit8<View>), dim3(groups), dim3(256), 0, s, v, ws.cells, ws.below, ws.first, order_in, nodes_out, order_out, ws.ctr, ws.queue,
                           ws.queue_cap);
    }
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_collapse_finish<View>), dim3(1), dim3(64), 0, s, ws.ctr, node_count_out, error_out);
    return hipGetLastError();
}

} // namespace collapse8
} // namespace rfwhip
