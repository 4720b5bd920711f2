// General edge lists on the GPU: the HIP executor and the C ABI of csrc/oard_general.h (see there).  Its own translation unit: nothing
// here touches the production path's kernels, tables or globals.
#include <hip/hip_runtime.h>

#include <new>

#include "oard_general.h"

namespace og = oard_general;

namespace {

template <class F>
__global__ __launch_bounds__(256) void k_general_stage(long long n, F f) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) f(i);
}

struct HipExec {
    hipStream_t st;
    template <class F>
    int run(long long n, const F& f) {
        if (n <= 0) return OARD_OK;
        const long long grid = (n + 255) / 256;
        if (grid > 0x7fffffffLL) return OARD_EINVAL;
        hipLaunchKernelGGL(k_general_stage<F>, dim3((unsigned)grid), dim3(256), 0, st, n, f);
        return hipGetLastError() == hipSuccess ? OARD_OK : OARD_EHIP;
    }
    int zero(void* p, size_t bytes) { return hipMemsetAsync(p, 0, bytes, st) == hipSuccess ? OARD_OK : OARD_EHIP; }
};

}  // namespace

struct oard_graph {
    og::GraphHost host;
    og::Graph dev;
    int32_t* block = nullptr;          // one device allocation behind every table of `dev`
    int device = 0;
    int complete = 0;                  // the edge SET is the complete graph per combined_mask value, no self loops, no duplicates
};

extern "C" {

static int graph_create_impl(const oard_config* c, const int64_t* cm, const int64_t* nfs, int64_t n_nodes, const int64_t* ei, int64_t n_edges,
                             oard_graph** out);
int oard_graph_create(const oard_config* c, const int64_t* cm, const int64_t* nfs, int64_t n_nodes, const int64_t* ei, int64_t n_edges,
                      oard_graph** out) {
    try {                                                 // (std::vector growth: no C++ exception may cross the C ABI)
        return graph_create_impl(c, cm, nfs, n_nodes, ei, n_edges, out);
    } catch (...) {
        return OARD_ENOMEM;
    }
}
static int graph_create_impl(const oard_config* c, const int64_t* cm, const int64_t* nfs, int64_t n_nodes, const int64_t* ei, int64_t n_edges,
                             oard_graph** out) {
    if (!c || !out || (n_nodes > 0 && (!cm || !nfs)) || (n_edges > 0 && !ei) || c->n_obj < 1 || c->n_obj > OARD_MAX_OBJECTS) return OARD_EINVAL;
    oard_graph* g = new (std::nothrow) oard_graph();
    if (!g) return OARD_ENOMEM;
    int rc = og::build_graph(c, cm, nfs, n_nodes, ei, n_edges, g->host);
    if (rc != OARD_OK) { delete g; return rc; }
    struct Guard { oard_graph* g; ~Guard() { if (g) { if (g->block) (void)hipFree(g->block); delete g; } } } guard{g};      // until *out owns it
    og::GraphHost& h = g->host;
    // complete per sample?  (what oard_topology_check_edge_index answers for the production path): every node has exactly the other members of
    // its combined_mask value as sources, once each
    {
        std::vector<int64_t> members;                                   // members per mask value
        int64_t maxv = -1;
        for (int64_t n = 0; n < h.N; ++n) maxv = std::max<int64_t>(maxv, h.node_tidx[n]);
        members.assign((size_t)(maxv + 1), 0);
        for (int64_t n = 0; n < h.N; ++n) ++members[h.node_tidx[n]];
        bool ok = true;
        std::vector<int32_t> seen(h.N, -1);
        for (int64_t n = 0; n < h.N && ok; ++n) {
            const int lo = h.in_ptr[n], hi = h.in_ptr[n + 1];
            if (hi - lo != members[h.node_tidx[n]] - 1) { ok = false; break; }
            for (int p = lo; p < hi; ++p) {
                const int i = h.ei0[h.in_list[p]];
                if (i == n || h.node_tidx[i] != h.node_tidx[n] || seen[i] == (int32_t)n) { ok = false; break; }
                seen[i] = (int32_t)n;
            }
        }
        g->complete = ok ? 1 : 0;
    }
    (void)hipGetDevice(&g->device);
    const std::vector<int32_t>* tabs[13] = {&h.ei0, &h.ei1, &h.in_ptr, &h.in_list, &h.out_ptr, &h.out_list, &h.sub, &h.node_obj, &h.node_row,
                                            &h.node_tidx, &h.node_grp, &h.grp_ptr, &h.grp_list};
    size_t off[14];
    off[0] = 0;
    for (int i = 0; i < 13; ++i) off[i + 1] = off[i] + ((tabs[i]->size() + 63) / 64) * 64;
    std::vector<int32_t> stage(off[13] ? off[13] : 64, 0);
    for (int i = 0; i < 13; ++i)
        if (!tabs[i]->empty()) memcpy(stage.data() + off[i], tabs[i]->data(), tabs[i]->size() * sizeof(int32_t));
    if (hipMalloc((void**)&g->block, stage.size() * sizeof(int32_t)) != hipSuccess) { g->block = nullptr; (void)hipGetLastError(); return OARD_EHIP; }
    if (hipMemcpy(g->block, stage.data(), stage.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) return OARD_EHIP;
    const int32_t* b = g->block;
    g->dev = og::Graph{h.N, h.E, h.G, b + off[0], b + off[1], b + off[2], b + off[3], b + off[4], b + off[5], b + off[6], b + off[7],
                       b + off[8], b + off[9], b + off[10], b + off[11], b + off[12]};
    guard.g = nullptr;
    *out = g;
    return OARD_OK;
}

void oard_graph_destroy(oard_graph* g) {
    if (!g) return;
    if (g->block) {
        int cur = 0;
        (void)hipGetDevice(&cur);
        if (cur != g->device) (void)hipSetDevice(g->device);
        (void)hipDeviceSynchronize();                     // a forward that still reads the tables may be in flight
        (void)hipFree(g->block);
        if (cur != g->device) (void)hipSetDevice(cur);
    }
    delete g;
}

int64_t oard_graph_num_nodes(const oard_graph* g) { return g ? g->host.N : 0; }
int64_t oard_graph_num_edges(const oard_graph* g) { return g ? g->host.E : 0; }
int oard_graph_is_complete(const oard_graph* g) { return g ? g->complete : 0; }
int64_t oard_graph_object_rows(const oard_graph* g, int k) {
    return (g && k >= 0 && k < (int)g->host.obj_rows.size()) ? g->host.obj_rows[k] : -1;
}

static size_t table_bytes(const oard_config* c) {       // parameter pointers + xh + out pointer tables at the head of the workspace
    return (((size_t)og::Params(c).count + 2 * OARD_MAX_OBJECTS) * sizeof(void*) + 255) & ~(size_t)255;
}

size_t oard_graph_workspace_bytes(const oard_config* c, const oard_graph* g) {
    if (!c || !g) return 0;
    return table_bytes(c) + og::carve(c, g->host.N, g->host.E, g->host.G, nullptr).bytes;
}

int oard_graph_forward(const oard_config* c, const oard_graph* g, const float* const* params, size_t n_params, const float* const* xh,
                       const float* t, int t_is_scalar, const float* cond, float* const* out, void* ws, size_t ws_bytes, int32_t* status,
                       oard_stream_t stream) {
    if (!c || !g || !params || !xh || !out || !ws) return OARD_EINVAL;
    const og::Params pi(c);
    if (n_params != (size_t)pi.count || c->pos_dim != 3 || c->hidden % 4 || c->hidden < 4 || c->num_radial < 1 || c->num_layers < 1) return OARD_EINVAL;
    if ((c->condition_time && !t) || (c->condition_nf > 0 && !cond)) return OARD_EINVAL;
    if (ws_bytes < oard_graph_workspace_bytes(c, g)) return OARD_ENOMEM;
    hipStream_t st = (hipStream_t)stream;
    // the pointer tables the stage kernels dereference: [params | xh | out] at the head of the caller's workspace (per call, so calls on
    // distinct workspaces do not share anything)
    const size_t np = (size_t)pi.count;
    constexpr size_t TAB_MAX = 1024;
    const void* tab[TAB_MAX] = {};
    const size_t n_tab = np + 2 * OARD_MAX_OBJECTS;
    if (n_tab > TAB_MAX || c->n_obj > OARD_MAX_OBJECTS) return OARD_EINVAL;
    for (size_t i = 0; i < np; ++i) tab[i] = params[i];
    for (int k = 0; k < c->n_obj; ++k) { tab[np + k] = xh[k]; tab[np + OARD_MAX_OBJECTS + k] = out[k]; }
    // pageable source: the runtime stages it before hipMemcpyAsync returns, `tab` may go out of scope afterwards
    if (hipMemcpyAsync(ws, tab, n_tab * sizeof(void*), hipMemcpyHostToDevice, st) != hipSuccess) return OARD_EHIP;
    const float* const* P_dev = (const float* const*)ws;
    const float* const* xh_dev = P_dev + np;
    float* const* out_dev = (float* const*)(P_dev + np + OARD_MAX_OBJECTS);
    const og::Workspace w = og::carve(c, g->host.N, g->host.E, g->host.G, (char*)ws + table_bytes(c));
    HipExec ex{st};
    return og::forward(ex, c, g->dev, params, P_dev, xh_dev, t, t_is_scalar, cond, out_dev, w, status);
}

}  // extern "C"
