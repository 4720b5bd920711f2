// General edge lists on the GPU: the HIP executor and the C ABI of csrc/oard_general.h (see there).  Its own translation unit: nothing
// here touches the production path's kernels, tables or globals.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>

#include "oard_general.h"

namespace og = oard_general;

namespace {

template <class F>
__global__ __launch_bounds__(256) void k_general_stage(long long n, F f) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) f(i);
}

// The dense layer (og::Gemm: gathered row segments x W^T, bias / SiLU / row scale / residual epilogue) on the float64 matrix pipe:
// v_mfma_f64_16x16x4_f64, A = 16 rows x 4 k of the input, B = 4 k x 16 outputs of W, both converted from the float32 they are stored in,
// the sum carried in float64 as og::Gemm carries it (the two differ in the ORDER of that sum only).  A wave owns RT x 16 rows and up to four
// 16-output tiles (the tiles of a layer dealt evenly to ceil(tiles / 4) waves: 196 outputs = 4 + 3 + 3 + 3); the four waves of a workgroup
// share their rows (L1), and the workgroups of one row block are consecutive on ONE XCD (shared L2).  A lane reads its operands as float4:
// lane (g, j) takes k = 16 s + 4 g ... + 3 of row j / output j, and component c of both serves MFMA step c - the instruction's k index is
// just a label, any assignment that is the same for A and B sums the same products.  No LDS, one 16-k step loaded ahead; the matrix pipe
// needs 64 cycles per instruction on gfx950 (float64 matrix peak = float64 vector peak), which is what hides the loads.
// Result layout: component r of lane (g, j) = row g + 4 r, output j (tools/micro/mfma_f64_layout.hip).
typedef double gd4 __attribute__((ext_vector_type(4)));
typedef float gf4 __attribute__((ext_vector_type(4)));

template <int RT>
__global__ __launch_bounds__(256, 3) void k_general_gemm_f64(og::Gemm p, int tiles, int groups, int colblocks, long long nblocks) {
    const long long per = (long long)gridDim.x / 8;                       // the grid is padded to a multiple of 8: block b runs on XCD b % 8
    const long long lin = (long long)(blockIdx.x % 8) * per + blockIdx.x / 8;
    if (lin >= nblocks) return;
    const long long rb = lin / colblocks;
    const int group = (int)(lin - rb * colblocks) * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // wave-uniform, and known to be
    if (group >= groups) return;                                           // (no barrier below: a wave may leave alone)
    const int base = tiles / groups, extra = tiles % groups;
    const int nt = base + (group < extra ? 1 : 0);
    const int t0 = group * base + (group < extra ? group : extra);
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
    const long long r0 = rb * (16 * RT);
    gd4 acc[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[rt][q] = gd4{0.0, 0.0, 0.0, 0.0};
    const float* wp[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int o = 16 * (t0 + (q < nt ? q : nt - 1)) + j;                      // a tile this wave does not own: its last one again (a cache hit;
        if (o >= p.nout) o = p.nout - 1;                                    // the number of loads in flight stays a constant of the loop)
        wp[q] = p.W + (size_t)o * p.ldw;
    }
    int koff = 0;
    for (int sgi = 0; sgi < p.nseg; ++sgi) {
        const og::Seg sg = p.seg[sgi];
        const float* xp[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            long long row = r0 + 16 * rt + j;
            if (row >= p.rows) row = p.rows - 1;
            const long long src = sg.idx ? (long long)sg.idx[row] : row;
            xp[rt] = sg.x + (size_t)src * sg.ld;
        }
        const int S = (sg.K + 15) / 16;
        gf4 a0[RT], b0[4], a1[RT], b1[4];
        const gf4 zero4 = {0.f, 0.f, 0.f, 0.f};
        // K % 4 == 0: a lane's float4 is inside the segment or outside it, never across.  Outside: read the segment's last float4 instead and
        // clear the INPUT half of the product (the weight read there belongs to this output and this segment)
        auto load = [&](int s, gf4* A, gf4* B) {
            int k = 16 * s + 4 * g;
            const bool valid = k < sg.K;
            if (!valid) k = sg.K - 4;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const gf4 v = *(const gf4*)(xp[rt] + k);
                A[rt] = valid ? v : zero4;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) B[q] = *(const gf4*)(wp[q] + koff + k);
        };
        auto mma = [&](const gf4* A, const gf4* B) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (q < nt) {
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt)
                            acc[rt][q] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)A[rt][c], (double)B[q][c], acc[rt][q], 0, 0, 0);
                    }
        };
        load(0, a0, b0);
        for (int s = 0; s < S; s += 2) {                                    // two named buffers: the step ahead is in flight under this step's MFMAs
            load(s + 1 < S ? s + 1 : S - 1, a1, b1);                        // (past the end: the last step again, unused - an unconditional load
            mma(a0, b0);                                                    //  keeps the count of loads in flight, and so the waits, exact)
            load(s + 2 < S ? s + 2 : S - 1, a0, b0);
            if (s + 1 < S) mma(a1, b1);
        }
        koff += sg.K;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int o = 16 * (t0 + q) + j;
        if (q >= nt || o >= p.nout) continue;
        const double bv = p.bias ? (double)p.bias[o] : 0.0;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long long row = r0 + 16 * rt + g + 4 * r;
                if (row >= p.rows) continue;
                float v = (float)(acc[rt][q][r] + bv);                    // the epilogue of og::Gemm, statement for statement
                if (p.act) v = og::silu_f(v);
                if (p.rowscale) v *= p.rowscale[row];
                float* y = p.Y + (size_t)row * p.ldy + o;
                if (p.mode == 1) v = *y + v;
                else if (p.mode == 2) v = p.resid[(size_t)row * p.ldr + o] + v;
                *y = v;
            }
    }
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

struct HipExec {
    hipStream_t st;
    bool matrix_pipe;                                   // false: every dense layer as og::Gemm on plain threads (OARD_GENERAL_GEMM=threads; the cross-check)
    template <class F>
    int run(long long n, const F& f) {
        if (n <= 0) return OARD_OK;
        const long long grid = (n + 255) / 256;
        if (grid > 0x7fffffffLL) return OARD_EINVAL;
        hipLaunchKernelGGL(k_general_stage<F>, dim3((unsigned)grid), dim3(256), 0, st, n, f);
        return hipGetLastError() == hipSuccess ? OARD_OK : OARD_EHIP;
    }
    int zero(void* p, size_t bytes) { return hipMemsetAsync(p, 0, bytes, st) == hipSuccess ? OARD_OK : OARD_EHIP; }
    int gemm(const og::Gemm& k) {
        // float4 operand reads: every segment a multiple of 4 columns on 16-byte rows (H, 2 H, 3 H, num_radial, the edge width; not the
        // 3-column frame input or an odd in_hidden: those few node-level layers stay on plain threads)
        bool ok = matrix_pipe && k.rows > 0 && k.nout > 0 && aligned16(k.W) && k.ldw % 4 == 0;
        for (int i = 0; i < k.nseg && ok; ++i) ok = k.seg[i].K > 0 && k.seg[i].K % 4 == 0 && k.seg[i].ld % 4 == 0 && aligned16(k.seg[i].x);
        if (!ok) return run(k.threads(), k);
        const int tiles = (k.nout + 15) / 16, groups = (tiles + 3) / 4, colblocks = (groups + 3) / 4;
        const int rt = k.rows >= 65536 ? 2 : 1;                          // edge-level layers: 32 rows per wave halve the weight reads
        const long long rowblocks = (k.rows + 16 * rt - 1) / (16 * rt), nblocks = rowblocks * colblocks, grid = (nblocks + 7) / 8 * 8;
        if (grid > 0x7fffffffLL) return OARD_EINVAL;
        if (rt == 2) hipLaunchKernelGGL(k_general_gemm_f64<2>, dim3((unsigned)grid), dim3(256), 0, st, k, tiles, groups, colblocks, nblocks);
        else hipLaunchKernelGGL(k_general_gemm_f64<1>, dim3((unsigned)grid), dim3(256), 0, st, k, tiles, groups, colblocks, nblocks);
        return hipGetLastError() == hipSuccess ? OARD_OK : OARD_EHIP;
    }
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
    const char* how = getenv("OARD_GENERAL_GEMM");
    HipExec ex{st, !(how && strcmp(how, "threads") == 0)};
    return og::forward(ex, c, g->dev, params, P_dev, xh_dev, t, t_is_scalar, cond, out_dev, w, status);
}

}  // extern "C"
