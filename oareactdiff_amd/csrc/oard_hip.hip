// oard_hip.hip — host side of liboard_hip.so: the C ABI declared in include/oard.h.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC oard_hip.hip -o liboard_hip.so
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include <mutex>
#include <chrono>
#include <utility>

#include "oard_kernels.h"
#include "oard_edge_v1.h"
#include "oard_edge_p.h"      // persistent form of the GCL throughput shape (round 5)
#include "oard_wgrad_t16.h"
#include "oard_edge_b3.h"     // split-precision (3 x bf16, fp32 accumulate) variant of the GCL edge kernel: optional (debug option gcl_b3)
#ifdef OARD_EXPERIMENTS
#include "oard_edge_fp.h"     // barrier-free variant of the GCL kernel: measured slower (profiles/round2_gcl_phase_study.txt), experiment builds only
#endif
#include "oard_node_v1.h"
#include "oard_edge_small.h"
#include "oard_edge_bwd.h"
#include "oard_node_bwd.h"
#include "oard_rows.h"
#include "oard_inst.h"        // the heavy kernel families are instantiated in their own translation units: `extern template` here

#define OARD_VERSION 2040      // round 6: + oard_graph_* (general edge lists), oard_library_stream
// The first-generation kernels (weights straight from L2, one wave per 16 nodes: gcl_variant / equi_variant / node_variant 0) are the
// A/B baseline of round 1 and a cross-check in tests/test_hip_parity.py::test_every_kernel_variant_is_parity_green; they are compiled
// into experiment builds (-DOARD_EXPERIMENTS) only - a product library refuses those variants (oard_debug_option returns OARD_EINVAL).
#ifdef OARD_EXPERIMENTS
static constexpr bool kV0 = true;
#else
static constexpr bool kV0 = false;
#endif

#define HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "liboard_hip: %s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    return OARD_EHIP; } } while (0)

// ------------------------------------------------------------------------------------------------
// per-family kernel timing with HIP events on the launch stream (bench.py's roofline leg)
// ------------------------------------------------------------------------------------------------
namespace {
enum Family { F_GCL_EDGE = 0, F_EQUI_EDGE, F_NODE, F_INIT, F_OTHER, F_GCL_BWD, F_EQUI_BWD, F_WGRAD, F_COUNT };
const char* kFamilyNames[F_COUNT] = {"gcl_edge", "equi_edge", "node", "init", "other", "gcl_edge_bwd", "equi_edge_bwd", "wgrad"};
struct TimingRec { hipEvent_t a, b; int fam; };
struct Timing {
    bool on = false;
    std::vector<TimingRec> recs;
    std::vector<hipEvent_t> pool;
    double total_ms[F_COUNT] = {0};
    long long launches[F_COUNT] = {0};
    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e; (void)hipEventCreate(&e); return e;
    }
    void flush() {
        for (auto& r : recs) {
            (void)hipEventSynchronize(r.b);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, r.a, r.b);
            total_ms[r.fam] += ms; launches[r.fam] += 1;
            pool.push_back(r.a); pool.push_back(r.b);
        }
        recs.clear();
    }
} g_timing;
int g_stop_after = 0;
int g_gcl_variant = 2;     // 0: v0 (weights straight from L2), 2: LDS-streamed 8-wave shape, 3: 4-wave shape, 6: latency kernel
int g_equi_variant = 2;    // 0: v0, 2: LDS-streamed 8-wave shape, 1: 4-wave shape, 4: latency kernel
int g_sequential = 0;       // 1: run the sub-batches one after the other on the caller stream (profiling)
int g_auto_tiny = 8;        // launches of <= 512 * g_auto_tiny edge tiles (all concurrent sub-batches together) use the
                            //    latency kernels of oard_edge_small.h (8 waves share 16 edges); 0 = never
int g_auto_small = 4;       // 1: small launches use the 4-wave workgroups (one wave per SIMD instead of two): a launch that cannot
                            //    fill the chip anyway finishes sooner when its waves do not share a SIMD (B <= 8 reactions: -25 %)
int g_npb = 0;              // nodes per workgroup of the node stages (0 = auto, see oard_topology_create)
int g_poison = 0;           // 1: fill the workspace with NaN bit patterns before every forward (tests: nothing may depend on its contents)
int g_parts = 0;            // sub-batches per topology (0 = auto: 4 for B >= 32, 2 for B >= 16, else 1)
int g_small_split = 128;     // EquiMessage latency kernel: launches of <= this many 16-edge tiles run one launch per dense stage
                            // (a workgroup = 16 edges x 8 output tiles); 0 = never
int g_wgrad_shapes = 3;       // bit k: workgroup shape kWglShapes[k] of the LDS-panel kernel may be chosen (A/B: debug option wgrad_shapes)
int g_wgrad_t16 = 256;       // workgroups per weight-gradient GEMM of the 16 x 16-tile kernel (oard_wgrad_t16.h; 0: never use it)
int g_wgrad_lds = 256;       // workgroups per weight-gradient GEMM of the LDS-panel kernel (0: always the per-wave-tile kernel k_wgrad)
int g_wgrad_wgs = 512;       // workgroups per weight-gradient GEMM (row chunks x task groups): one round of 2 x 4 waves per CU
                            // (measured per training step: 384 -> 38.1 ms, 512 -> 30.7, 768 -> 36.0, 1024 -> 33.2, 2048 -> 38.1)
int g_gcl_persist = 1;      // the fp32 GCL throughput shape as a persistent workgroup (oard_edge_p.h); 0: one 128-edge tile per workgroup
int g_gcl_grid = 0;         // workgroups per launch of the persistent kernel: 0 = auto (one per CU when the launch has the chip to itself,
                            //    half of them when sub-batches run concurrently), > 0 = that many
int g_equi_skip = 1;        // EquiMessage runs the inner edges inside the cutoff only (ActList; exact: it is zero on the others).  0: every inner row
int g_gcl_skip = 1;         // skip S1 (first layer) / S3 (last layer) on inter-object edges
int g_node_variant = 1;     // 0: one wave per 16 nodes, 1: 8 waves per 16 nodes with LDS-resident activations
// Arithmetic of the two MFMA edge stages: oard_config::precision (OARD_PREC_* bits, include/oard.h) - a property of the CALL, read
// when the weights are packed (the bf16 streams are only built for the bits set) and when the stages are launched.  There is no
// process-wide precision state: two modules with different precision may run from two threads / on two streams.
int g_skip_families = 0;    // timing experiments only (results are garbage): bit f set = launches of family f are dropped

struct ScopedLaunch {
    hipStream_t st; hipEvent_t a; int fam; bool on;
    ScopedLaunch(int f, hipStream_t s) : st(s), fam(f), on(g_timing.on) {
        if (on) { a = g_timing.get(); (void)hipEventRecord(a, st); }
    }
    ~ScopedLaunch() {
        if (on) { hipEvent_t b = g_timing.get(); (void)hipEventRecord(b, st); g_timing.recs.push_back({a, b, fam}); }
    }
};
#define LAUNCH(fam, kern, grid, block, stream, ...) do { ScopedLaunch sl_(fam, stream); if (!((g_skip_families >> (fam)) & 1)) \
    hipLaunchKernelGGL(kern, dim3((unsigned)(grid)), dim3(block), 0, stream, __VA_ARGS__); } while (0)
#define LAUNCH2(fam, kern, gx, gy, block, stream, ...) do { ScopedLaunch sl_(fam, stream); if (!((g_skip_families >> (fam)) & 1)) \
    hipLaunchKernelGGL(kern, dim3((unsigned)(gx), (unsigned)(gy)), dim3(block), 0, stream, __VA_ARGS__); } while (0)

#ifndef OARD_DIMS_LIST      // production (train_ts1x.py:43-56) + the two test widths of the reference's unit tests / goldens
#define OARD_DIMS_LIST X(196, 96) X(32, 8) X(32, 32)
#endif

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline long long cdiv(long long a, long long b) { return (a + b - 1) / b; }

bool config_ok(const oard_config* c) {
    if (!c) return false;
    bool dims = false;                      // (hidden_channels, num_radial) pairs this build instantiates (OARD_DIMS_LIST)
#define X(h, r) dims = dims || (c->hidden == (h) && c->num_radial == (r));
    OARD_DIMS_LIST
#undef X
    if (!dims) return false;
    if (c->num_layers < 1 || c->num_layers > OARD_MAX_LAYERS) return false;
    if (c->in_hidden < 1 || c->in_hidden > 16) return false;
    if (c->n_obj < 1 || c->n_obj > OARD_MAX_OBJECTS) return false;
    if (c->pos_dim != 3 || (c->reflect_equiv != 0 && c->reflect_equiv != 1)) return false;
    if (c->precision & ~(OARD_PREC_GCL_BF16X3 | OARD_PREC_EQUI_BF16X3 | OARD_PREC_TRAIN_BF16X3)) return false;
    const int emb = c->in_hidden - (c->condition_time ? 1 : 0) - (c->condition_nf > 0 ? c->condition_nf : 0);
    if (emb < 1) return false;
    for (int k = 0; k < c->n_obj; ++k) {
        const int d = c->node_nf[k] - 3;
        if (d < 1 || d > 16) return false;
        const int a = c->enc_alias[k];
        if (a < 0 || a >= c->n_obj || c->node_nf[a] != c->node_nf[k]) return false;
    }
    return true;
}
int embed_dim(const oard_config* c) {
    return c->in_hidden - (c->condition_time ? 1 : 0) - (c->condition_nf > 0 ? c->condition_nf : 0);
}

// ------------------------------------------------------------------------------------------------
// packed blob layout
// ------------------------------------------------------------------------------------------------
PackOff make_layout(const oard_config* c) {
    const RDims d(c->hidden, c->num_radial);
    PackOff po;
    memset(&po, 0, sizeof(po));
    size_t cur = 0;
    auto take = [&](size_t n) { size_t o = cur; cur = align_up(cur + n, 64); return o; };
    auto mat = [&](int MT, int KB) { return take((size_t)MT * KB * 256); };
    const int emb = embed_dim(c);
    po.emb = mat(d.HT, 1); po.emb_b = take(d.HP);
    po.nbemb = mat(d.HT, 1); po.nbemb_b = take(d.HP);
    po.s2v = mat(d.HT, d.HT); po.s2v_b = take(d.HP);
    po.rl0 = mat(d.HT, d.RB); po.rl0_b = take(d.HP);
    po.rl2 = mat(d.HT, d.HT); po.rl2_b = take(d.HP);
    po.lin3 = take((size_t)d.H4 * 5 + 1);
    po.pe0 = take((size_t)d.H2 * 3);
    po.pe1 = mat(d.HT, d.PB);
    po.embout = mat(1, d.HT); po.embout_b = take(16);
    po.v1p = mat(d.HT, d.HT); po.v2p = take(d.HP);
    po.un0 = mat(d.HT, 2 * d.HT); po.un0_b = take(d.HP);
    po.un2 = mat(1, d.HT); po.un2_b = take(16);
    po.c0row = take(d.WP);
    po.u0 = take(d.HP);
    po.rbf_means = take(d.RP); po.rbf_betas = take(d.RP);
    for (int k = 0; k < c->n_obj; ++k) {
        const int dd = c->node_nf[k] - 3;
        po.enc[k] = take((size_t)2 * dd * dd + 2 * dd + (size_t)emb * 2 * dd + emb);
        po.dec[k] = take((size_t)2 * dd * emb + 2 * dd + (size_t)dd * 2 * dd + dd);
    }
    for (int l = 0; l < c->num_layers; ++l) {
        LayerOff& lo = po.layer[l];
        lo.ln_g_w = take(d.HP); lo.ln_g_b = take(d.HP);
        lo.W1a = mat(d.HT, d.HT); lo.b1 = take(d.HP);
        lo.W1b = mat(d.HT, d.HT);
        lo.W1c = mat(d.HT, d.WB);
        lo.W2 = mat(d.HT, d.HT); lo.b2 = take(d.HP);
        lo.watt = take(d.HP); lo.batt = take(1);
        lo.W3 = mat(d.WB, d.HT); lo.b3 = take(d.WP);
        lo.nm0 = mat(d.HT, 2 * d.HT); lo.nm0b = take(d.HP);
        lo.nm1 = mat(d.HT, d.HT); lo.nm1b = take(d.HP);
        lo.ln_q_w = take(d.HP); lo.ln_q_b = take(d.HP);
        lo.xp0 = mat(d.HT, d.HT);
        lo.xp2 = mat(3 * d.HT, d.HT);
        lo.dp0 = mat(d.D1T, d.WB); lo.dp0b = take(d.D1P);
        lo.dp2 = mat(3 * d.HT, d.D1T); lo.dp2b = take(3 * d.HP);
        lo.rbfp = mat(3 * d.HT, d.RB);
        lo.vp = mat(2 * d.HT, d.HT);
        lo.xv0 = mat(d.HT, 2 * d.HT);
        lo.xv2 = mat(3 * d.HT, d.HT);
        lo.l3u = take(593);
        lo.l3t = take(L3T_FLOATS);
        lo.gcl_stream = take((size_t)(d.WB * d.HT + (d.HT + 1) * (d.HT + 1) + d.WB * (d.HT + 1)) * 256);
        lo.equi_stream = take((size_t)(d.WB * d.D1T + 3 * d.HT * (1 + d.D1T + d.RB)) * 256);
        {   // GclB3Stream<D>::CHUNKS
            const int nbh = (d.HT + 1) / 2, nbw = (d.WB + 1) / 2;
            lo.gcl_b3 = take((size_t)(nbw * 3 * d.HT + (d.HT + 1) * (1 + 3 * nbh) + d.WB * (1 + 3 * nbh)) * 256);
            // EquiB3Stream<D>::CHUNKS
            lo.equi_b3 = take((size_t)(nbw * 3 * d.D1T + ((d.RB + 1) / 2 + (d.D1T + 1) / 2) * 9 * d.HT) * 256);
        }
    }
    po.total = cur;
    po.signed_scal = c->reflect_equiv ? 0 : 1;
    for (int l = 0; l < c->num_layers; ++l) po.layer[l].xcross = c->reflect_equiv ? 0 : 1;
    return po;
}

// canonical parameter order == oareactdiff_amd/spec.py:state_spec == reference state_dict()
struct ParamIdx {
    int emb_w, emb_b, embout_w, embout_b, means, betas, nbemb_w, nbemb_b, s2v_w, s2v_b, rl0_w, rl0_b, rl2_w, rl2_b,
        lin30_w, lin30_b, lin32_w, lin32_b, pe0_w, pe1_w;
    int gcl0, msg0, upd0, out0, enc0, dec0, count;
    explicit ParamIdx(const oard_config* c) {
        int i = 0;
        emb_w = i++; emb_b = i++; embout_w = i++; embout_b = i++; means = i++; betas = i++;
        nbemb_w = i++; nbemb_b = i++; s2v_w = i++; s2v_b = i++; rl0_w = i++; rl0_b = i++; rl2_w = i++; rl2_b = i++;
        lin30_w = i++; lin30_b = i++; lin32_w = i++; lin32_b = i++; pe0_w = i++; pe1_w = i++;
        i += 2;                                   // distance_embedding (unused in forward)
        gcl0 = i; i += 14 * c->num_layers;
        msg0 = i; i += 9 * c->num_layers;
        upd0 = i; i += 9 * c->num_layers;
        i += 2;                                   // last_layer (unused in forward)
        out0 = i; i += 6;
        enc0 = i; i += 4 * c->n_obj;
        dec0 = i; i += 4 * c->n_obj;
        count = i;
    }
};

struct Packer {
    const float* const* p; float* blob; hipStream_t st;
    std::vector<GenJob> jobs;
    long long blocks = 0;
    void push(GenJob j, size_t work) {
        j.block0 = blocks;
        blocks += cdiv((long long)std::max<size_t>(work, 1), 256);
        jobs.push_back(j);
    }
    void matrix(int src, int src_ld, int col_off, int msl, int msp, int ms, int ksl, int ksp, int ks, int MT, int KB,
                size_t dst, size_t tstride = 0, size_t bstride = 256, int perm_ht = 0, int transpose = 0, int tail_compact = 0,
                int rows4 = 0) {
        if (tstride == 0) tstride = (size_t)KB * 256;
        GenJob j;
        memset(&j, 0, sizeof(j));
        j.type = 0;
        j.m = PackJob{p[src], src_ld, col_off, msl, msp, ms, ksl, ksp, ks, MT, KB, dst, tstride, bstride, perm_ht, transpose, rows4, tail_compact};
        push(j, (size_t)MT * KB * 256);
    }
    // natural (unsectioned) matrix [M][K] taken from columns [col_off, col_off+K) of a [M][src_ld] tensor
    void nat(int src, int src_ld, int col_off, int M, int K, int MT, int KB, size_t dst) {
        matrix(src, src_ld, col_off, M, MT * 16, 1, K, KB * 16, 1, MT, KB, dst);
    }
    void vec(int src, int sect_len, int sect_pad, int sects, int n_dst, size_t dst) {
        GenJob j;
        memset(&j, 0, sizeof(j));
        j.type = 1; j.src = src >= 0 ? p[src] : nullptr; j.dst = dst; j.sect_len = sect_len; j.sect_pad = sect_pad; j.sects = sects; j.n = n_dst;
        push(j, (size_t)n_dst);
    }
    void bias_chunks(int src, int sect_len, int sect_pad, int sects, int n_tiles, size_t dst, size_t tstride,
                     int perm_ht = 0) {
        GenJob j;
        memset(&j, 0, sizeof(j));
        j.type = 2; j.src = p[src]; j.dst = dst; j.sect_len = sect_len; j.sect_pad = sect_pad; j.sects = sects; j.n = n_tiles * 256;
        j.tstride = tstride; j.perm_ht = perm_ht;
        push(j, (size_t)n_tiles * 256);
    }
    void raw(int src, int n, size_t dst) {
        GenJob j;
        memset(&j, 0, sizeof(j));
        j.type = 3; j.src = p[src]; j.dst = dst; j.n = n;
        push(j, (size_t)n);
    }
    // one launch for everything recorded so far.  The job table lives on the device, cached per blob address: it only changes when the
    // parameter tensors move (it is compared with the cached host copy on every call, a few hundred KiB of memcmp)
    int flush() {
        if (jobs.empty()) return OARD_OK;
        struct Cached { std::vector<GenJob> host; GenJob* dev = nullptr; size_t cap = 0; };
        static std::mutex mu;
        static std::map<std::pair<const void*, int>, Cached> cache;
        int devid = 0;
        (void)hipGetDevice(&devid);
        const GenJob* table = nullptr;
        {
            std::lock_guard<std::mutex> lk(mu);
            Cached& c = cache[{(const void*)blob, devid}];
            const bool same = c.host.size() == jobs.size() && memcmp(c.host.data(), jobs.data(), jobs.size() * sizeof(GenJob)) == 0;
            if (!same) {
                HIP_TRY(hipStreamSynchronize(st));            // a launch that still reads the old table may be in flight
                if (c.cap < jobs.size()) {
                    if (c.dev) (void)hipFree(c.dev);
                    HIP_TRY(hipMalloc((void**)&c.dev, jobs.size() * sizeof(GenJob)));
                    c.cap = jobs.size();
                }
                HIP_TRY(hipMemcpy(c.dev, jobs.data(), jobs.size() * sizeof(GenJob), hipMemcpyHostToDevice));
                c.host = jobs;
            }
            table = c.dev;
        }
        hipLaunchKernelGGL(k_pack_all, dim3((unsigned)blocks), dim3(256), 0, st, table, (int)jobs.size(), blob);
        jobs.clear();
        blocks = 0;
        return OARD_OK;
    }
};
// scratch of the stage-split latency path (activations between the stage launches), set per forward call / sub-batch
#define OARD_SMALL_MAX_EDGES (512 * 16 * 16)      // topologies up to 131 072 edges get the scratch buffer
struct SmallScratch { float* a; float* b; };
static thread_local SmallScratch t_small = {nullptr, nullptr};
static WsOff make_ws(const oard_config* c, const TopoDev& td) {
    const RDims d(c->hidden, c->num_radial);
    const size_t N = td.N, E = td.E + 1, A = td.A + 1;   // + the spare row of the padding columns
    WsOff w;
    size_t cur = 0;
    auto take = [&](size_t bytes) { size_t o = cur; cur = align_up(cur + bytes, 256); return o; };
    w.pos = take(N * 3 * 4); w.pf64 = take(N * 3 * 8); w.pf32 = take(N * 3 * 4); w.x1 = take(N * 3 * 4);
    w.pp0 = take(N * 4); w.labels = take(N * 4); w.hin = take(N * 16 * 4);
    w.zemb = take(N * d.HP * 4); w.nb = take(N * d.HP * 4); w.s = take(N * d.HP * 4); w.s1 = take(N * d.HP * 4);
    w.ne1 = take(N * 3 * d.HP * 4); w.xh = take(N * d.HP * 4); w.P = take(N * d.HP * 4); w.Q = take(N * d.HP * 4);
    w.xq = take(N * 3 * d.HP * 4); w.vec = take(N * 3 * d.HP * 4); w.vec2 = take(N * 3 * d.HP * 4); w.v2buf = take(N * 3 * d.HP * 4);
    w.sc0 = take(N * d.HP * 4); w.vdot = take(N * d.HP * 4);
    w.geo = take(A * GEO_STRIDE * 4); w.d64 = take(A * 8); w.rbuf = take(A * d.RP * 4);
    w.ew = take(E * d.WP * 4); w.mbuf = take(E * d.HP * 4);
    w.xmsg = take(A * d.HP * 4); w.vmsg = take(A * 3 * d.HP * 4);
    w.dpos = take(N * 3 * 4); w.hout = take(N * 16 * 4);
    // scratch of the stage-split latency edge kernels: only topologies small enough to ever take that path (the launch-shape
    // thresholds are run-time options, so the launch re-checks that the buffers exist)
    w.al_rows = take(A * 4); w.al_src = take(A * 4); w.al_pre = take(A * 4); w.al_cnt = take((size_t)cdiv((long long)td.A, 256) * 4 + 4); w.al_n = take(4);
    w.d1s = take((size_t)cdiv((long long)td.A, 128) * 8 * d.D1T * 1024);
    w.small_a = w.small_b = 0;
    if (E <= (size_t)OARD_SMALL_MAX_EDGES) {
        w.small_a = take(A * d.D1P * 4);
    }
    w.total = cur;
    return w;
}



template <class K>
int set_lds(K kernel, size_t bytes) {
    HIP_TRY(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return OARD_OK;
}

// the dynamic-LDS limit is a per-device function attribute: track it per (call site, device)
#define LAUNCH_LDS(fam, kern, grid, block, lds, stream, ...) do { \
    static bool attr_done_[64] = {}; \
    int dev_ = 0; (void)hipGetDevice(&dev_); dev_ &= 63; \
    if (!attr_done_[dev_]) { int rc_ = set_lds(kern, lds); if (rc_ != OARD_OK) return rc_; attr_done_[dev_] = true; } \
    ScopedLaunch sl_(fam, stream); if (!((g_skip_families >> (fam)) & 1)) \
    hipLaunchKernelGGL(kern, dim3((unsigned)(grid)), dim3(block), lds, stream, __VA_ARGS__); } while (0)

#define GCL_CASE(id, WV_, GP_) case id: { \
        LAUNCH_LDS(F_GCL_EDGE, (k_gcl_edge_v1<D, WV_, GP_, S1, S3, false>), cdiv(r1 - r0, 16 * WV_), WV_ * 64, \
                   (GclStream<D, GP_>::LDS_BYTES), st, tp, stream, P, Q, u0, c0, r0, r1, ew_in, ew_out, mbuf, GclTape{}); return OARD_OK; }
// the throughput shape: 8 waves x 16 edges, two waves per SIMD, three LDS slabs (barrier inside the phase)
#define GCL_RING3(TRAIN_, tape_) do { \
        LAUNCH_LDS(F_GCL_EDGE, (k_gcl_edge_v1<D, 8, 2, S1, S3, TRAIN_, 2, 3>), cdiv(r1 - r0, 16 * 8), 8 * 64, \
                   (GclStream<D, 2>::LDS_BYTES / 2 * 3), st, tp, stream, P, Q, u0, c0, r0, r1, ew_in, ew_out, mbuf, tape_); return OARD_OK; } while (0)
static int device_cus() {
    static int cus[64] = {};
    int d = 0;
    (void)hipGetDevice(&d);
    d &= 63;
    if (!cus[d]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n < 1) { (void)hipGetLastError(); n = 256; }
        cus[d] = n;
    }
    return cus[d];
}
template <class D, bool S1, bool S3>
int launch_gcl_v1s(int prec, int variant, int conc, const TopoDev& tp, const float* wb, const LayerOff& lo, const float* stream, const float* P, const float* Q, const float* u0,
                   const float* c0, long long r0, long long r1, const float* ew_in, float* ew_out, float* mbuf, const GclTape* tape, hipStream_t st) {
    if (r1 <= r0) return OARD_OK;
    if (tape && (prec & OARD_PREC_TRAIN_BF16X3)) {    // training-mode forward in split precision (optional)
        LAUNCH_LDS(F_GCL_EDGE, (k_gcl_edge_b3<D, S1, S3, true>), cdiv(r1 - r0, 16 * 8), 8 * 64, (GclB3Stream<D>::LDS_BYTES), st, tp,
                   wb + lo.gcl_b3, P, Q, u0, c0, r0, r1, ew_in, ew_out, mbuf, *tape);
        return OARD_OK;
    }
    if (tape) {                  // training-mode forward: one shape (8 waves x 16 edges), pre-activations stored
#ifdef OARD_EXPERIMENTS
        if (variant == 7) {
            LAUNCH_LDS(F_GCL_EDGE, (k_gcl_edge_v1<D, 8, 2, S1, S3, true>), cdiv(r1 - r0, 16 * 8), 8 * 64,
                       (GclStream<D, 2>::LDS_BYTES), st, tp, stream, P, Q, u0, c0, r0, r1, ew_in, ew_out, mbuf, *tape);
            return OARD_OK;
        }
#endif
        if (g_gcl_persist) {     // the training-mode forward is one sub-batch: persistent workgroups, balanced last round (oard_edge_p.h)
            const long long nhalf = cdiv(cdiv(r1 - r0, 16), 4);
            const long long want = g_gcl_grid > 0 ? g_gcl_grid : (g_gcl_grid < 0 ? cdiv(nhalf, 2LL * -g_gcl_grid) : device_cus());
            LAUNCH_LDS(F_GCL_EDGE, (k_gcl_edge_p<D, S1, S3, true>), std::min(nhalf, want), 8 * 64, (GclStream<D, 2>::LDS_BYTES / 2 * 3), st,
                       tp, stream, P, Q, u0, c0, r0, r1, ew_in, ew_out, mbuf, *tape);
            return OARD_OK;
        }
        GCL_RING3(true, *tape);
    }
    if (variant == 2 && cdiv(r1 - r0, 16) * conc <= 1024LL * g_auto_small) variant = 3;
    if ((variant == 2 || variant == 3) && cdiv(r1 - r0, 16) * conc <= 512LL * g_auto_tiny) variant = 6;
    if (variant == 6) {          // latency kernel: 8 waves share 16 edges
        LAUNCH_LDS(F_GCL_EDGE, (k_gcl_edge_small<D, 8, S1, S3>), cdiv(r1 - r0, 16), 512, (GclSmall<D>::LDS_BYTES), st,
                   tp, wb, lo, P, Q, u0, c0, r0, r1, ew_out, mbuf);
        return OARD_OK;
    }
    switch (variant) {
        case 2:
            if (prec & OARD_PREC_GCL_BF16X3) {
                LAUNCH_LDS(F_GCL_EDGE, (k_gcl_edge_b3<D, S1, S3>), cdiv(r1 - r0, 16 * 8), 8 * 64, (GclB3Stream<D>::LDS_BYTES), st, tp,
                           wb + lo.gcl_b3, P, Q, u0, c0, r0, r1, ew_in, ew_out, mbuf, GclTape{});
                return OARD_OK;
            }
            // A launch that has the chip to itself runs as persistent workgroups over half-tiles (oard_edge_p.h; same arithmetic, bit-identical
            // rows): its last round is balanced (isolated 9.01 -> 8.62 ms per B = 64 step).  Concurrent sub-batches keep one tile per workgroup:
            // there the other streams' kernels fill the tail, and workgroups that hold a CU for several tiles make THEM wait (measured:
            // 17.93 ms per step with tiles, 18.05 - 18.3 with 64 ... 256 persistent workgroups per launch; gcl_persist = 2 forces it).
            if (g_gcl_persist > 1 || (g_gcl_persist == 1 && conc == 1)) {
                const long long nhalf = cdiv(cdiv(r1 - r0, 16), 4);
                // gcl_grid < 0: -k = k rounds (128-row tiles) per workgroup
                const long long want = g_gcl_grid > 0 ? g_gcl_grid : (g_gcl_grid < 0 ? cdiv(nhalf, 2LL * -g_gcl_grid) : (conc > 1 ? std::max(1, device_cus() / 2) : device_cus()));
                LAUNCH_LDS(F_GCL_EDGE, (k_gcl_edge_p<D, S1, S3>), std::min(nhalf, want), 8 * 64, (GclStream<D, 2>::LDS_BYTES / 2 * 3), st,
                           tp, stream, P, Q, u0, c0, r0, r1, ew_in, ew_out, mbuf, GclTape{});
                return OARD_OK;
            }
            GCL_RING3(false, GclTape{});
        GCL_CASE(3, 4, 2)      // 4 waves x 16 edges (two workgroups per CU): small launches
#ifdef OARD_EXPERIMENTS        // A/B shapes, only in experiment builds (tools/ab_gcl.sh)
        GCL_CASE(7, 8, 2)      // the throughput shape with two slabs (barrier at the phase start; round 1 / 2)
        case 4: {              // flag pipeline: no workgroup barriers, ring of 4 slabs (oard_edge_fp.h)
            LAUNCH_LDS(F_GCL_EDGE, (k_gcl_edge_fp<D, 8, 2, S1, S3, false>), cdiv(r1 - r0, 16 * 8), 8 * 64,
                       (GclRing<D, 2, 4>::LDS_BYTES), st, tp, stream, P, Q, u0, c0, r0, r1, ew_in, ew_out, mbuf, GclTape{}); return OARD_OK; }
        case 5: {              // the same with a ring of 3 slabs (waves 0..3 at most one phase ahead)
            LAUNCH_LDS(F_GCL_EDGE, (k_gcl_edge_fp<D, 8, 2, S1, S3, false, 3>), cdiv(r1 - r0, 16 * 8), 8 * 64,
                       (GclRing<D, 2, 3>::LDS_BYTES), st, tp, stream, P, Q, u0, c0, r0, r1, ew_in, ew_out, mbuf, GclTape{}); return OARD_OK; }
        GCL_CASE(8, 8, 3)
        GCL_CASE(9, 8, 4)
        case 10: { LAUNCH_LDS(F_GCL_EDGE, (k_gcl_edge_v1<D, 12, 1, S1, S3, false, 3>), cdiv(r1 - r0, 16 * 12), 12 * 64,
                              (GclStream<D, 1>::LDS_BYTES), st, tp, stream, P, Q, u0, c0, r0, r1, ew_in, ew_out, mbuf, GclTape{}); return OARD_OK; }
        case 11: { LAUNCH_LDS(F_GCL_EDGE, (k_gcl_edge_v1<D, 12, 2, S1, S3, false, 3>), cdiv(r1 - r0, 16 * 12), 12 * 64,
                              (GclStream<D, 2>::LDS_BYTES), st, tp, stream, P, Q, u0, c0, r0, r1, ew_in, ew_out, mbuf, GclTape{}); return OARD_OK; }
        case 12: { LAUNCH_LDS(F_GCL_EDGE, (k_gcl_edge_v1<D, 4, 1, S1, S3, false, 3>), cdiv(r1 - r0, 16 * 4), 4 * 64,
                              (GclStream<D, 1>::LDS_BYTES), st, tp, stream, P, Q, u0, c0, r0, r1, ew_in, ew_out, mbuf, GclTape{}); return OARD_OK; }
#endif
        default: return OARD_EINVAL;
    }
}
// one GCL edge pass of layer l: inner edges always run every stage; inter-object edges skip S1 in the first
// layer (constant initial state) and S3 in the last (their updated state is never read)
template <class D>
int launch_gcl_v1(int prec, int variant, int conc, const TopoDev& tp, const float* wb, const LayerOff& lo, const float* stream, const float* P, const float* Q, const float* u0,
                  const float* c0, bool first, bool last, const float* ew_in, float* ew_out, float* mbuf, const GclTape* tape, hipStream_t st) {
    const long long A = tp.A, E = tp.E;
    int rc;
    const bool skip = g_gcl_skip || tape;
    if (!skip || (!first && !last)) return launch_gcl_v1s<D, true, true>(prec, variant, conc, tp, wb, lo, stream, P, Q, u0, c0, 0, E, ew_in, ew_out, mbuf, tape, st);
    rc = launch_gcl_v1s<D, true, true>(prec, variant, conc, tp, wb, lo, stream, P, Q, u0, c0, 0, A, ew_in, ew_out, mbuf, tape, st);
    if (rc != OARD_OK) return rc;
    if (first && last) return launch_gcl_v1s<D, false, false>(prec, variant, conc, tp, wb, lo, stream, P, Q, u0, c0, A, E, ew_in, ew_out, mbuf, tape, st);
    if (first) return launch_gcl_v1s<D, false, true>(prec, variant, conc, tp, wb, lo, stream, P, Q, u0, c0, A, E, ew_in, ew_out, mbuf, tape, st);
    return launch_gcl_v1s<D, true, false>(prec, variant, conc, tp, wb, lo, stream, P, Q, u0, c0, A, E, ew_in, ew_out, mbuf, tape, st);
}
#define EQUI_CASE(id, WV_) case id: { \
        LAUNCH_LDS(F_EQUI_EDGE, (k_equi_edge_v1<D, WV_, false>), cdiv(tp.A, 16 * WV_), WV_ * 64, (EquiStream<D>::LDS_BYTES), st, \
                   tp, stream, dp0b, ew, rbuf, qbuf, nullptr, nullptr, al); return OARD_OK; }
template <class D>
int launch_equi_v1(int prec, int variant, int conc, const TopoDev& tp, const float* wb, const LayerOff& lo, const float* stream, const float* dp0b, const float* ew,
                   const float* rbuf, float* qbuf, float* zd1, float* cd, hipStream_t st, float* d1s = nullptr, ActList al = ActList{}) {
    // (the latency kernels run every inner row: their q is zero on the rows outside the cutoff, which the node stage does not read when
    // it walks the list; the throughput kernels of both precisions take the list)
    if (zd1 && (prec & OARD_PREC_TRAIN_BF16X3) && d1s) {      // training-mode forward in split precision (optional)
        LAUNCH_LDS(F_EQUI_EDGE, (k_equi_edge_b3<D, true>), cdiv(tp.A, 16 * 8), 8 * 64, (EquiB3Stream<D>::LDS_BYTES), st, tp, wb + lo.equi_b3,
                   dp0b, wb + lo.dp2b, ew, rbuf, qbuf, d1s, zd1, cd, ActList{});
        return OARD_OK;
    }
    if (zd1) {                   // training-mode forward
        LAUNCH_LDS(F_EQUI_EDGE, (k_equi_edge_v1<D, 8, true>), cdiv(tp.A, 16 * 8), 8 * 64, (EquiStream<D>::LDS_BYTES), st,
                   tp, stream, dp0b, ew, rbuf, qbuf, zd1, cd, ActList{});
        return OARD_OK;
    }
    if (variant == 2 && cdiv(tp.A, 16) * conc <= 512LL * g_auto_small) variant = 1;
    if ((variant == 2 || variant == 1) && cdiv(tp.A, 16) * conc <= 512LL * g_auto_tiny) variant = 4;
    if (variant == 4) {          // latency kernel: 8 waves share 16 edges
        if (t_small.a && cdiv(tp.A, 16) * conc <= g_small_split) {   // stage-split latency path: 2 launches, few tiles only
            const long long tiles = cdiv(tp.A, 16);
            LAUNCH_LDS(F_EQUI_EDGE, (k_equi_small_s1<D, 8>), tiles * cdiv(D::D1T, 8), 512, (size_t)D::WB * 1024, st,
                       tp, wb, lo, ew, t_small.a);
            LAUNCH_LDS(F_EQUI_EDGE, (k_equi_small_s2<D, 8>), tiles * cdiv(3 * D::HT, 8), 512, (size_t)(D::D1T + D::RB) * 1024, st,
                       tp, wb, lo, (const float*)t_small.a, rbuf, qbuf);
            return OARD_OK;
        }
        LAUNCH_LDS(F_EQUI_EDGE, (k_equi_edge_small<D, 8>), cdiv(tp.A, 16), 512, (EquiSmall<D>::LDS_BYTES), st,
                   tp, wb, lo, ew, rbuf, qbuf);
        return OARD_OK;
    }
    if (variant == 2 && (prec & OARD_PREC_EQUI_BF16X3) && d1s) {      // split precision (oard_edge_b3.h): the throughput shape only
        LAUNCH_LDS(F_EQUI_EDGE, (k_equi_edge_b3<D>), cdiv(tp.A, 16 * 8), 8 * 64, (EquiB3Stream<D>::LDS_BYTES), st, tp, wb + lo.equi_b3,
                   dp0b, wb + lo.dp2b, ew, rbuf, qbuf, d1s, nullptr, nullptr, al);
        return OARD_OK;
    }
    switch (variant) {
        EQUI_CASE(1, 4)
        EQUI_CASE(2, 8)
        default: return OARD_EINVAL;
    }
}

// ------------------------------------------------------------------------------------------------
// training tape: what a training-mode forward keeps for the backward pass (byte offsets)
// ------------------------------------------------------------------------------------------------
struct TapeOff {
    size_t hin, geo, rbuf, pp0, x1;                                  // init-stage constants
    size_t s_in[OARD_MAX_LAYERS + 1], vec_in[OARD_MAX_LAYERS + 1];  // node state at the start of layer l (index L: final)
    size_t agg[OARD_MAX_LAYERS], s_mid[OARD_MAX_LAYERS];           // mean message per node; s after the GCL node update
    size_t s_a[OARD_MAX_LAYERS], vec_a[OARD_MAX_LAYERS];           // s, vec after the EquiMessage aggregation (before EquiUpdate)
    size_t ew[OARD_MAX_LAYERS + 1];                                  // edge state entering layer l (index L: final)
    size_t z1[OARD_MAX_LAYERS], z2[OARD_MAX_LAYERS], att[OARD_MAX_LAYERS], z3[OARD_MAX_LAYERS];
    size_t zd1[OARD_MAX_LAYERS], cd[OARD_MAX_LAYERS];
    size_t total;
};
static TapeOff make_tape(const oard_config* c, const TopoDev& td) {
    const RDims d(c->hidden, c->num_radial);
    const size_t N = td.N, E = td.E + 1, A = td.A + 1;
    TapeOff t;
    memset(&t, 0, sizeof(t));
    size_t cur = 0;
    auto take = [&](size_t bytes) { size_t o = cur; cur = align_up(cur + bytes, 256); return o; };
    t.hin = take(N * 16 * 4); t.geo = take(A * GEO_STRIDE * 4); t.rbuf = take(A * d.RP * 4); t.pp0 = take(N * 4); t.x1 = take(N * 3 * 4);
    for (int l = 0; l <= c->num_layers; ++l) {
        t.s_in[l] = take(N * d.HP * 4); t.vec_in[l] = take(N * 3 * d.HP * 4); t.ew[l] = take(E * d.WP * 4);
        if (l == c->num_layers) break;
        t.agg[l] = take(N * d.HP * 4); t.s_mid[l] = take(N * d.HP * 4);
        t.s_a[l] = take(N * d.HP * 4); t.vec_a[l] = take(N * 3 * d.HP * 4);
        t.z1[l] = take(E * d.HP * 4); t.z2[l] = take(E * d.HP * 4); t.att[l] = take(E * 4); t.z3[l] = take(E * d.WP * 4);
        t.zd1[l] = take(A * d.D1P * 4); t.cd[l] = take(A * 3 * d.HP * 4);
    }
    t.total = cur;
    return t;
}

// `tape` != NULL: training-mode forward (TapeOff layout): every layer's input edge state lives in its own tape
// buffer (the update is out of place), the edge kernels store their pre-activations, and the node state at the
// layer boundaries is written straight into its tape slots (no copies).  One launch shape, layer-0 / last-layer shortcuts always on.
template <class D>
static int forward_impl(const oard_config* c, const TopoPart* topo, const float* wb, const float* const* xh,
                        const float* t, int t_scalar, const float* cond, float* const* out, char* ws, char* tape,
                        int* status, hipStream_t st) {
    const TopoDev& tp = topo->d;
    const PackOff po = make_layout(c);
    const WsOff w = make_ws(c, tp);
    const bool train = tape != nullptr;
    const TapeOff to = train ? make_tape(c, tp) : TapeOff{};
    const int emb = embed_dim(c);
    float* pos = (float*)(ws + w.pos); double* pf64 = (double*)(ws + w.pf64); float* pf32 = (float*)(ws + w.pf32);
    // training: the buffers whose final contents the tape keeps (hin, geo, rbuf, pp0, x1, the vector state at the layer boundaries) ARE
    // their tape slots - the kernels write them in place of a device-to-device copy afterwards (round 4: 12 of the 25 copies per step)
    auto slot = [&](size_t ws_off, size_t tape_off) -> float* { return train ? (float*)(tape + tape_off) : (float*)(ws + ws_off); };
    float* x1 = slot(w.x1, to.x1); float* pp0 = slot(w.pp0, to.pp0); int* labels = (int*)(ws + w.labels);
    float* hin = slot(w.hin, to.hin); float* zemb = (float*)(ws + w.zemb); float* nb = (float*)(ws + w.nb);
    float* s = (float*)(ws + w.s); float* s1 = (float*)(ws + w.s1); float* ne1 = (float*)(ws + w.ne1);
    float* xhb = (float*)(ws + w.xh); float* P = (float*)(ws + w.P); float* Q = (float*)(ws + w.Q);
    float* xq = (float*)(ws + w.xq); float* vec = slot(w.vec, to.vec_in[0]); float* v2buf = (float*)(ws + w.v2buf);
    float* scal = (float*)(ws + w.sc0); float* vdot = (float*)(ws + w.vdot);
    float* geo = slot(w.geo, to.geo); double* d64 = (double*)(ws + w.d64); float* rbuf = slot(w.rbuf, to.rbuf);
    float* mbuf = (float*)(ws + w.mbuf); float* xmsg = (float*)(ws + w.xmsg);
    float* vmsg = (float*)(ws + w.vmsg); float* dpos = (float*)(ws + w.dpos); float* hout = (float*)(ws + w.hout);
    t_small = (w.small_a && !train) ? SmallScratch{(float*)(ws + w.small_a), (float*)(ws + w.small_b)} : SmallScratch{nullptr, nullptr};
    // edge state entering layer l (inference: one buffer updated in place)
    auto ew_at = [&](int l) -> float* { return train ? (float*)(tape + to.ew[l]) : (float*)(ws + w.ew); };
    float* ew = ew_at(0);
    // scalar node state entering layer l / after the GCL node update of layer l (inference: one buffer updated in place)
    auto s_at = [&](int l) -> float* { return train ? (float*)(tape + to.s_in[l]) : s; };
    auto s_mid_at = [&](int l) -> float* { return train ? (float*)(tape + to.s_mid[l]) : s; };

    ObjPtrs op;
    memset(&op, 0, sizeof(op));
    for (int k = 0; k < c->n_obj; ++k) {
        op.xh[k] = xh[k]; op.out[k] = out[k]; op.node_nf[k] = c->node_nf[k]; op.enc[k] = po.enc[k]; op.dec[k] = po.dec[k];
    }
    const long long N = tp.N, E = tp.E, A = tp.A;
    const unsigned gN = (unsigned)cdiv(N, 64), gE = (unsigned)cdiv(E, 64), gA = (unsigned)cdiv(A, 64);
    const double cutoff = (double)c->cutoff;
    const int gcl_variant = train ? 2 : g_gcl_variant, equi_variant = train ? 2 : g_equi_variant;
    const int node_variant = train ? 1 : g_node_variant;
    const bool gcl_skip = train || g_gcl_skip;
    const int stop_after = train ? 0 : g_stop_after;
    // the inner rows inside the cutoff, compacted per call: EquiMessage (edge kernel and node-side gather) runs those only.  Inference with
    // the streamed edge kernels and the v1 node stages; the training-mode forward tapes every row for its backward pass.
    const bool use_al = !train && g_equi_skip && A > 0 && equi_variant != 0 && node_variant == 1;
    int* al_cnt = (int*)(ws + w.al_cnt);
    const ActList al = use_al ? ActList{(const int*)(ws + w.al_rows), (const int*)(ws + w.al_src), (const int*)(ws + w.al_pre), (const int*)(ws + w.al_n)}
                              : ActList{nullptr, nullptr, nullptr, nullptr};
    // a call that builds no list says so in the workspace: oard_active_inner_edges reports what the LAST call on it did, not what an
    // earlier inference call left behind (0xff bytes = -1)
    if (!use_al) HIP_TRY(hipMemsetAsync(ws + w.al_n, 0xff, sizeof(int), st));
    LAUNCH(F_OTHER, k_prep, cdiv(N, 128), 128, st, tp, op, wb, pos, hin, t, t_scalar, cond,
           c->condition_nf > 0 ? c->condition_nf : 0, c->condition_time, emb);
    LAUNCH(F_INIT, k_geom, tp.n_groups, 64, st, tp, (const float*)pos, cutoff, pf64, pf32, x1, pp0, labels);
    // inter-object rows start as the constant row; with the layer-0 shortcut nobody reads them before layer 0
    // writes them, so they are only materialised for the unspecialised paths and for the debug tap
    if (!(gcl_skip && gcl_variant != 0) || stop_after == 1)
        LAUNCH(F_INIT, k_fill_edges, std::min<long long>(cdiv((E - A + 1) * (D::WP / 4), 256), 8192), 256, st,
               wb + po.c0row, ew + (size_t)A * D::WP, E - A + 1, D::WP);
    if (A > 0) {
        LAUNCH(F_INIT, k_edge_geo, cdiv(A, 256), 256, st, tp, (const float*)pos, (const double*)pf64, cutoff, geo, d64, use_al ? al_cnt : nullptr);
        if (use_al) LAUNCH(F_INIT, k_active_list, cdiv(A, 256), 256, st, tp, (const float*)geo, (const int*)al_cnt, (int*)al.rows, (int*)al.src, (int*)al.pre, (int*)al.n);
        LAUNCH(F_INIT, k_rbf, cdiv(A * D::RP, 256), 256, st, tp, (const double*)d64, (const float*)geo,
               wb + po.rbf_means, wb + po.rbf_betas, cutoff, rbuf, ew, D::R, D::RP, D::H, D::WP);
    }
    LAUNCH(F_INIT, (k_node_embed<D>), gN, 256, st, tp, wb, po, (const float*)hin, zemb, nb);
    if (A > 0) LAUNCH(F_INIT, (k_radial_lin<D>), gA, 256, st, tp, wb, po, (const float*)rbuf, (const float*)geo, ew);
    // waves per workgroup of the v1 node stages: one per hidden tile (13 at H = 196), so that every wave owns
    // exactly one tile of each H-wide layer
    constexpr int NW = D::HT <= 16 ? D::HT : 8;
    const unsigned gNb = (unsigned)cdiv(N, tp.npb);
    if (node_variant >= 1) {
        LAUNCH(F_INIT, (k_neighbor_v1<D, NW>), gNb, NW * 64, st, tp, wb, po, (const float*)zemb, (const float*)nb, (const float*)ew, s_at(0), s1);
        LAUNCH(F_INIT, (k_s2v_agg_v1<D, NW>), gNb, NW * 64, st, tp, (const float*)s1, (const float*)ew, (const float*)geo, ne1);
    } else if constexpr (kV0) {
        LAUNCH(F_INIT, (k_neighbor<D>), gN, 256, st, tp, wb, po, (const float*)zemb, (const float*)nb, (const float*)ew, s_at(0), s1);
        LAUNCH(F_INIT, (k_s2v_agg<D>), gN, 256, st, tp, (const float*)s1, (const float*)ew, (const float*)geo, ne1);
    } else return OARD_EINVAL;
    // small launches: one workgroup per (64 edges, hidden tile) instead of per 64 edges
    if (A > 0) LAUNCH2(F_INIT, (k_scalarize<D>), gA, (gA * topo->conc <= 2048 ? D::HT : 1), 256, st, tp, wb, po, (const float*)ne1, (const float*)geo, ew);
    HIP_TRY(hipMemsetAsync(vec, 0, (size_t)N * 3 * D::HP * sizeof(float), st));
    if (stop_after == 1) return OARD_OK;

    float* vec2 = (float*)(ws + w.vec2);
    float* vcur = vec;          // holds the current vec; v1 ping-pongs between vec and vec2
    float* vnext = vec2;
    for (int l = 0; l < c->num_layers; ++l) {
        const LayerOff lo = po.layer[l];
        const bool nv1 = node_variant == 1 && equi_variant != 0;
        const unsigned gN16 = (unsigned)cdiv(N, tp.npb);
        float* ew_in = ew_at(l);
        float* ew_out = ew_at(l + 1);
        if (train) vnext = (float*)(tape + to.vec_in[l + 1]);  // vcur == tape slot vec_in[l]

        if (nv1) LAUNCH(F_NODE, (k_node_pre_v1<D, NW>), gN16, NW * 64, st, tp, wb, po, lo, (const float*)s_at(l), (const float*)pp0, xhb, P, Q);
        else if constexpr (kV0) LAUNCH(F_NODE, (k_node_pre<D>), gN, 256, st, tp, wb, po, lo, (const float*)s, (const float*)pp0, xhb, P, Q);
        else return OARD_EINVAL;
        if (E > 0) {
            if (gcl_variant == 0) {
                if constexpr (kV0) LAUNCH(F_GCL_EDGE, (k_gcl_edge<D>), gE, 256, st, tp, wb, lo, (const float*)P, (const float*)Q, ew_in, mbuf);
                else return OARD_EINVAL;
            } else {
                GclTape gt{};
                if (train) gt = GclTape{(float*)(tape + to.z1[l]), (float*)(tape + to.z2[l]), (float*)(tape + to.att[l]), (float*)(tape + to.z3[l])};
                int rc = launch_gcl_v1<D>(c->precision, gcl_variant, topo->conc, tp, wb, lo, wb + lo.gcl_stream, P, Q, wb + po.u0, wb + po.c0row, l == 0,
                                          l == c->num_layers - 1, ew_in, ew_out, mbuf, train ? &gt : nullptr, st);
                if (rc != OARD_OK) return rc;
            }
        }
        const bool rows = tp.npb <= 4;                       // small batches: gathers walk the rows with the wave's columns (row_lanes)
        if (nv1 && rows) LAUNCH(F_NODE, (k_gcl_node_v1<D, NW, true>), gN16, NW * 64, st, tp, wb, lo, (const float*)xhb, (const float*)mbuf, s_mid_at(l), xq,
                                train ? (float*)(tape + to.agg[l]) : nullptr);
        else if (nv1) LAUNCH(F_NODE, (k_gcl_node_v1<D, NW, false>), gN16, NW * 64, st, tp, wb, lo, (const float*)xhb, (const float*)mbuf, s_mid_at(l), xq,
                             train ? (float*)(tape + to.agg[l]) : nullptr);
        else if constexpr (kV0) LAUNCH(F_NODE, (k_gcl_node<D>), gN, 256, st, tp, wb, lo, (const float*)xhb, (const float*)mbuf, s, xq);
        else return OARD_EINVAL;
        if (stop_after == 100 + 10 * l + 1) { topo->vec_final = (size_t)((char*)vcur - ws); return OARD_OK; }
        if (equi_variant == 0) {
            if constexpr (kV0) {
                if (A > 0) LAUNCH(F_EQUI_EDGE, (k_equi_edge<D>), gA, 256, st, tp, wb, lo, (const float*)ew_out, (const float*)rbuf,
                                  (const float*)geo, (const float*)xq, (const float*)vcur, xmsg, vmsg);
                LAUNCH(F_NODE, (k_equi_agg<D>), gN, 256, st, tp, wb, lo, (const float*)xmsg, (const float*)vmsg, (const float*)x1,
                       s, vcur, v2buf, scal, vdot);
            } else return OARD_EINVAL;
        } else {
            if (A > 0) {
                int rc = launch_equi_v1<D>(c->precision, equi_variant, topo->conc, tp, wb, lo, wb + lo.equi_stream, wb + lo.dp0b, ew_out, rbuf, vmsg,
                                           train ? (float*)(tape + to.zd1[l]) : nullptr, train ? (float*)(tape + to.cd[l]) : nullptr, st,
                                           (float*)(ws + w.d1s), al);
                if (rc != OARD_OK) return rc;
            }
#define EQUI_NODE_V1(ROWS_, XC_) LAUNCH(F_NODE, (k_equi_node_v1<D, NW, ROWS_, XC_>), gN16, NW * 64, st, tp, wb, lo, (const float*)vmsg, \
                       (const float*)xq, (const float*)geo, (const float*)x1, (const float*)s_mid_at(l), s_at(l + 1), (const float*)vcur, vnext, \
                       train ? (float*)(tape + to.s_a[l]) : nullptr, train ? (float*)(tape + to.vec_a[l]) : nullptr, al)
            if (nv1 && rows) {
                if (lo.xcross) EQUI_NODE_V1(true, true); else EQUI_NODE_V1(true, false);
            } else if (nv1) {
                if (lo.xcross) EQUI_NODE_V1(false, true); else EQUI_NODE_V1(false, false);
            } else if constexpr (kV0) {
                LAUNCH(F_NODE, (k_equi_agg_v1<D>), gN, 256, st, tp, wb, lo, (const float*)vmsg, (const float*)xq,
                       (const float*)geo, (const float*)x1, s, (const float*)vcur, vnext, v2buf, scal, vdot);
            } else return OARD_EINVAL;
            std::swap(vcur, vnext);
        }
        if constexpr (kV0) {
            if (!nv1) LAUNCH(F_NODE, (k_equi_upd<D>), gN, 256, st, tp, wb, lo, (const float*)scal, (const float*)vdot,
                             (const float*)v2buf, s, vcur);
        }
        if (stop_after == 100 + 10 * l + 2) { topo->vec_final = (size_t)((char*)vcur - ws); return OARD_OK; }
    }
    if (!train) topo->vec_final = (size_t)((char*)vcur - ws);
    if (train) {
        // The backward edge kernels run 128-row tiles whose padding columns read the SPARE row (index E / A) of every taped array and take
        // part in sums over a wave's columns with a zero cotangent: 0 x finite = 0, but 0 x NaN is not.  The forward's own padding columns
        // write those rows only when a launch HAS padding columns, with whatever the spare row of its input held (the tape is grow-only,
        // uninitialised memory) - finite so far by the accident of the layer-0 launch over the inter-object rows, and not at all once the
        // persistent kernel deals 64-row half-tiles (E % 128 == 64: poisoned-tape test, round 5).  So the spare rows are zeroed here, behind
        // every kernel that may have written them.
        ZeroRows z;
        z.count = 0;
        auto add = [&](size_t off, size_t row, int ld) {
            if (z.count == 64) { LAUNCH(F_OTHER, k_zero_rows, 64, 256, st, z); z.count = 0; }
            z.p[z.count] = (float*)(tape + off) + row * (size_t)ld; z.n[z.count] = ld; ++z.count;
        };
        for (int l = 0; l <= c->num_layers; ++l) {
            add(to.ew[l], (size_t)E, D::WP);
            if (l == c->num_layers) break;
            add(to.z1[l], (size_t)E, D::HP); add(to.z2[l], (size_t)E, D::HP); add(to.att[l], (size_t)E, 1); add(to.z3[l], (size_t)E, D::WP);
            add(to.zd1[l], (size_t)A, D::D1P); add(to.cd[l], (size_t)A, 3 * D::HP);
        }
        add(to.geo, (size_t)A, GEO_STRIDE); add(to.rbuf, (size_t)A, D::RP);
        if (z.count > 0) LAUNCH(F_OTHER, k_zero_rows, z.count, 256, st, z);
    }
    if (node_variant >= 1)
        LAUNCH(F_NODE, (k_out_v1<D, NW>), gNb, NW * 64, st, tp, wb, po, (const float*)s_at(c->num_layers), (const float*)vcur, dpos, hout, status);
    else if constexpr (kV0)
        LAUNCH(F_NODE, (k_out<D>), gN, 256, st, tp, wb, po, (const float*)s, (const float*)vcur, dpos, hout, status);
    else return OARD_EINVAL;
    LAUNCH(F_OTHER, k_post, cdiv(N, 128), 128, st, tp, op, wb, (const float*)dpos, (const float*)hout, emb);
    HIP_TRY(hipGetLastError());
    return OARD_OK;
}

// One instantiation of every kernel per (hidden_channels, num_radial) pair of OARD_DIMS_LIST (oareactdiff_amd/build.py:
// environment OARD_DIMS="196x96,32x8,32x32,..."; a checkpoint with other widths is a rebuild, not a code change).
template <class F> static int dispatch_dims(const oard_config* c, F&& f) {
#define X(h, r) if (c->hidden == (h) && c->num_radial == (r)) { f(Dims<(h), (r)>{}); return 0; }
    OARD_DIMS_LIST
#undef X
    return OARD_EINVAL;
}
#define DISPATCH_DIMS(c, CALL)                                                                         \
    do {                                                                                               \
        if (dispatch_dims((c), [&](auto d_) { using D = decltype(d_); CALL; }) != 0) return OARD_EINVAL; \
    } while (0)

// ---- transposed weight streams of the backward edge kernels -----------------------------------------------------
struct BwdLayerOff { size_t gcl, equi, watt; };
struct BwdOff { BwdLayerOff layer[OARD_MAX_LAYERS]; size_t total; };
static BwdOff make_bwd_layout(const oard_config* c) {
    const RDims d(c->hidden, c->num_radial);
    BwdOff b;
    memset(&b, 0, sizeof(b));
    size_t cur = 0;
    auto take = [&](size_t n) { size_t o = cur; cur = align_up(cur + n, 64); return o; };
    for (int l = 0; l < c->num_layers; ++l) {
        b.layer[l].gcl = take((size_t)(2 * d.WB * d.HT + d.HT * d.HT) * 256);
        b.layer[l].equi = take((size_t)(3 * d.HT * d.D1T + d.WB * d.D1T) * 256);
        b.layer[l].watt = take(d.HP);
    }
    b.total = cur;
    return b;
}
template <class D>
static int gcl_backward_impl(const oard_config* c, const TopoDev& tp, const float* pb, const BwdLayerOff& bl, int layer,
                             const char* tape, const TapeOff& to, const float* dagg, float* dew, float* dz3, float* mout,
                             float* dz2, float* da, float* dz1, hipStream_t st, float* gate_part = nullptr, long long* gate_rows = nullptr) {
    GclBwdArgs a;
    a.gate_part = gate_part;
    a.z1 = (const float*)(tape + to.z1[layer]); a.z2 = (const float*)(tape + to.z2[layer]);
    a.att = (const float*)(tape + to.att[layer]); a.z3 = (const float*)(tape + to.z3[layer]);
    a.dagg = dagg; a.watt = pb + bl.watt; a.dew = dew; a.dz3 = dz3; a.mout = mout; a.dz2 = dz2; a.da = da; a.dz1 = dz1;
    const float* stream = pb + bl.gcl;
    const bool last = layer == c->num_layers - 1;
    const long long r_full = last ? tp.A : tp.E;       // rows whose forward ran S3
    long long rows = 0;                                 // per-wave rows of gate_part written so far
    if (r_full > 0) {
        LAUNCH_LDS(F_GCL_BWD, (k_gcl_edge_bwd<D, 8, 2, true>), cdiv(r_full, 128), 512, (GclBwdStream<D, 2>::LDS_BYTES), st,
                   tp, stream, 0LL, r_full, a);
        rows += cdiv(r_full, 128) * 8;
    }
    if (last && tp.E > tp.A) {
        if (a.gate_part) a.gate_part += (size_t)rows * D::HP;
        LAUNCH_LDS(F_GCL_BWD, (k_gcl_edge_bwd<D, 8, 2, false>), cdiv(tp.E - tp.A, 128), 512, (GclBwdStream<D, 2>::LDS_BYTES), st,
                   tp, stream, tp.A, tp.E, a);
        rows += cdiv(tp.E - tp.A, 128) * 8;
    }
    if (gate_rows) *gate_rows = rows;
    HIP_TRY(hipGetLastError());
    return OARD_OK;
}

template <class D>
static int equi_backward_impl(const TopoDev& tp, const float* stream, const float* dcd, const float* zd1, float* dew,
                              float* dzd1, hipStream_t st) {
    LAUNCH_LDS(F_EQUI_BWD, (k_equi_edge_bwd<D, 8>), cdiv(tp.A, 128), 512, (EquiBwdStream<D>::LDS_BYTES), st, tp, stream, dcd,
               zd1, dew, dzd1);
    HIP_TRY(hipGetLastError());
    return OARD_OK;
}

#ifndef OARD_SCAL_BWD_WAVES
#define OARD_SCAL_BWD_WAVES 2
#endif
template <class D>
static int scalarize_backward_impl(const oard_config* c, const TopoDev& tp, const float* wb, const char* tape, const TapeOff& to,
                                   const float* ne1, int ld, const float* dew, float* dne1, float* part, hipStream_t st) {
    const PackOff po = make_layout(c);
    constexpr int NW = OARD_SCAL_BWD_WAVES;        // 2 HT units of (channel tile, side) per node, dealt to the waves
    if (po.signed_scal)
        LAUNCH(F_INIT, (k_scalarize_bwd<D, NW, true>), tp.N, NW * 64, st, tp, wb + po.lin3, ne1, ld, (const float*)(tape + to.geo), dew, dne1, part);
    else
        LAUNCH(F_INIT, (k_scalarize_bwd<D, NW, false>), tp.N, NW * 64, st, tp, wb + po.lin3, ne1, ld, (const float*)(tape + to.geo), dew, dne1, part);
    HIP_TRY(hipGetLastError());
    return OARD_OK;
}

template <class D>
static int equi_msg_backward_impl(const oard_config* c, const TopoDev& tp, const char* tape, const TapeOff& to, int layer, const float* xq, const float* cr,
                                  const float* gs, const float* gv, float* dcd, float* dcr, float* dxq, float* dvec, hipStream_t st) {
    const Strided3 xq3{xq, 3 * D::H, D::H}, vec3{(const float*)(tape + to.vec_in[layer]), 3 * D::HP, D::HP}, cr3{cr, 3 * D::H, D::H},
        gv3{gv, 3 * D::H, D::H};
    LAUNCH(F_NODE, (k_equi_msg_bwd<D>), tp.N, 256, st, tp, (const float*)(tape + to.geo), xq3, vec3, (const float*)(tape + to.cd[layer]), cr3,
           gs, D::H, gv3, dcd, dcr, dxq, dvec, D::H, c->reflect_equiv ? 0 : 1);
    HIP_TRY(hipGetLastError());
    return OARD_OK;
}

// plan of the weight-gradient GEMM: a pure function of the shape, so that the summation order (and with it the result,
// bit for bit) does not depend on anything else.  Q (16-wide tiles, no padding waste) is the narrower operand; when that is
// dY the kernel produces the transposed product.
struct WgradPlan { int transposed, NT, nPB, nQG, PP, QP, gy; long long rpc; int n_chunks; int lds, PBW, QGW; };
// workgroup shapes of the LDS-panel kernel: NT tiles per Q group, PBW P blocks x QGW Q groups = the 8 tasks of a workgroup
struct WglShape { int NT, PBW, QGW; };
static const WglShape kWglShapes[2] = {{7, 4, 2}, {8, 4, 2}};
static WgradPlan wgrad_plan(int ncY, int ncX, long long rows) {
    WgradPlan p;
    p.transposed = ncX > ncY ? 1 : 0;
    const int ncP = p.transposed ? ncX : ncY, ncQ = p.transposed ? ncY : ncX;
    const int nQT = (int)cdiv(ncQ, 16);
    p.NT = (cdiv(nQT, 7) * 7 <= cdiv(nQT, 8) * 8) ? 7 : 8;
    p.nPB = (int)cdiv(ncP, 64); p.nQG = (int)cdiv(nQT, p.NT);
    p.PP = p.nPB * 64; p.QP = p.nQG * p.NT * 16;
    p.gy = (int)cdiv((long long)p.nPB * p.nQG, 4);            // workgroups (4 waves = 4 (P block, Q group) tasks) per row chunk
    long long want = std::max<long long>(1, g_wgrad_wgs / p.gy);
    want = std::min(want, std::max<long long>(1, cdiv(rows, rows < 16384 ? 128 : 64)));      // short contractions: fewer, longer chunks (the partials dominate)
    p.rpc = align_up((size_t)cdiv(std::max<long long>(rows, 1), want), 4 * OARD_WG_PD);
    p.n_chunks = (int)cdiv(std::max<long long>(rows, 1), p.rpc);
    // long contractions with at least two Q groups: the LDS-panel kernel (k_wgrad_lds), one 8-wave workgroup per CU and round; a
    // workgroup owns PBW P blocks x QGW Q groups of a row chunk.  Of the shapes that fill >= 85 % of their wave slots with tasks the
    // one that executes the fewest padded columns (P slots x 64) x (Q slots x 16 NT) is taken: 684 x 196 -> NT 7 (768 x 224);
    // 684 x 588 -> NT 7 (768 x 672: 0.83 ms; NT 8, fewer padded Q tiles but a sixth, empty group slot: 768 x 768, 0.92 ms);
    // 588 x 588 (83 % of the slots) stays on the per-wave kernel (0.86 ms; here 0.93).  Workgroups of 2 P blocks x 4 Q groups with
    // NT 5 / 6 were measured too (684 x 588: 0.83 ms, 588 x 588: 0.96 ms): no better, not instantiated.
    p.lds = 0; p.PBW = 4; p.QGW = 2;
    if (g_wgrad_lds > 0 && rows >= 16384) {
        long long best = -1;
        for (int k = 0; k < 2; ++k) {
            if (!((g_wgrad_shapes >> k) & 1)) continue;
            const WglShape& c = kWglShapes[k];
            const long long nQG = cdiv(nQT, c.NT);
            if (nQG < 2) continue;
            const long long sP = cdiv(p.nPB, c.PBW) * c.PBW, sQ = cdiv(nQG, c.QGW) * c.QGW;
            if ((long long)p.nPB * nQG * 20 < sP * sQ * 17) continue;     // < 85 % of the wave slots would hold a task
            const long long cost = sP * 64 * sQ * c.NT * 16;
            if (best < 0 || cost < best) { best = cost; p.lds = 1; p.NT = c.NT; p.PBW = c.PBW; p.QGW = c.QGW; p.nQG = (int)nQG; }
        }
        if (p.lds) p.QP = p.nQG * p.NT * 16;
    }
    if (p.lds) {
        const long long tiles = cdiv(p.nPB, p.PBW) * cdiv(p.nQG, p.QGW);
        long long w2 = std::max<long long>(1, g_wgrad_lds / tiles);
        w2 = std::min(w2, std::max<long long>(1, cdiv(rows, 4 * WGL_ROWS)));
        p.rpc = align_up((size_t)cdiv(rows, w2), WGL_ROWS);
        p.n_chunks = (int)cdiv(rows, p.rpc);
        p.gy = (int)tiles;
    }
    return p;
}
}  // namespace

extern "C" {

int oard_version(void) { return OARD_VERSION; }
int oard_supported(const oard_config* cfg) { return config_ok(cfg) ? OARD_OK : OARD_EINVAL; }
size_t oard_param_count(const oard_config* cfg) { return config_ok(cfg) ? (size_t)ParamIdx(cfg).count : 0; }
size_t oard_packed_bytes(const oard_config* cfg) { return config_ok(cfg) ? make_layout(cfg).total * sizeof(float) : 0; }

int oard_pack_weights(const oard_config* c, const float* const* params, size_t n_params, void* packed,
                      size_t packed_bytes, oard_stream_t stream) {
    if (!config_ok(c) || !params || !packed) return OARD_EINVAL;
    const ParamIdx pi(c);
    if (n_params != (size_t)pi.count) return OARD_EINVAL;
    const PackOff po = make_layout(c);
    if (packed_bytes < po.total * sizeof(float)) return OARD_ENOMEM;
    const RDims d(c->hidden, c->num_radial);
    const int H = d.H, R = d.R, W = d.W, C = c->in_hidden, emb = embed_dim(c);
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(packed, 0, po.total * sizeof(float), st));
    Packer pk{params, (float*)packed, st, {}, 0};

    pk.nat(pi.emb_w, C, 0, H, C, d.HT, 1, po.emb);          pk.vec(pi.emb_b, H, d.HP, 1, d.HP, po.emb_b);
    pk.nat(pi.nbemb_w, C, 0, H, C, d.HT, 1, po.nbemb);      pk.vec(pi.nbemb_b, H, d.HP, 1, d.HP, po.nbemb_b);
    pk.nat(pi.s2v_w, H, 0, H, H, d.HT, d.HT, po.s2v);       pk.vec(pi.s2v_b, H, d.HP, 1, d.HP, po.s2v_b);
    pk.nat(pi.rl0_w, R, 0, H, R, d.HT, d.RB, po.rl0);       pk.vec(pi.rl0_b, H, d.HP, 1, d.HP, po.rl0_b);
    pk.nat(pi.rl2_w, H, 0, H, H, d.HT, d.HT, po.rl2);       pk.vec(pi.rl2_b, H, d.HP, 1, d.HP, po.rl2_b);
    pk.raw(pi.lin30_w, d.H4 * 3, po.lin3);
    pk.raw(pi.lin30_b, d.H4, po.lin3 + d.H4 * 3);
    pk.raw(pi.lin32_w, d.H4, po.lin3 + d.H4 * 4);
    pk.raw(pi.lin32_b, 1, po.lin3 + d.H4 * 5);
    pk.raw(pi.pe0_w, d.H2 * 3, po.pe0);
    pk.nat(pi.pe1_w, d.H2, 0, H, d.H2, d.HT, d.PB, po.pe1);
    pk.nat(pi.embout_w, H, 0, C, H, 1, d.HT, po.embout);    pk.vec(pi.embout_b, C, 16, 1, 16, po.embout_b);
    pk.raw(pi.means, R, po.rbf_means);                      pk.raw(pi.betas, R, po.rbf_betas);
    const int o = pi.out0;                                   // vec1_proj, vec2_proj, update_net.0 (w,b), update_net.2 (w,b)
    pk.nat(o + 0, H, 0, H, H, d.HT, d.HT, po.v1p);
    pk.vec(o + 1, H, d.HP, 1, d.HP, po.v2p);
    pk.matrix(o + 2, 2 * H, 0, H, d.HP, 1, H, d.HP, 2, d.HT, 2 * d.HT, po.un0);   pk.vec(o + 3, H, d.HP, 1, d.HP, po.un0_b);
    pk.nat(o + 4, H, 0, 2, H, 1, d.HT, po.un2);                                  pk.vec(o + 5, 2, 16, 1, 16, po.un2_b);
    for (int k = 0; k < c->n_obj; ++k) {
        const int a = c->enc_alias[k], dd = c->node_nf[k] - 3;
        const int e = pi.enc0 + 4 * a, q = pi.dec0 + 4 * a;
        size_t off = po.enc[k];
        pk.raw(e + 0, 2 * dd * dd, off); off += 2 * dd * dd;
        pk.raw(e + 1, 2 * dd, off); off += 2 * dd;
        pk.raw(e + 2, emb * 2 * dd, off); off += emb * 2 * dd;
        pk.raw(e + 3, emb, off);
        off = po.dec[k];
        pk.raw(q + 0, 2 * dd * emb, off); off += 2 * dd * emb;
        pk.raw(q + 1, 2 * dd, off); off += 2 * dd;
        pk.raw(q + 2, dd * 2 * dd, off); off += dd * 2 * dd;
        pk.raw(q + 3, dd, off);
    }
    for (int l = 0; l < c->num_layers; ++l) {
        const LayerOff& lo = po.layer[l];
        const int g = pi.gcl0 + 14 * l;   // edge_mlp.0 w,b | edge_mlp.1 w,b | node_mlp.0 w,b | node_mlp.1 w,b | edge_out w,b | att w,b | ln w,b
        const int ld0 = 2 * H + W;
        pk.nat(g + 0, ld0, 0, H, H, d.HT, d.HT, lo.W1a);     pk.vec(g + 1, H, d.HP, 1, d.HP, lo.b1);
        pk.nat(g + 0, ld0, H, H, H, d.HT, d.HT, lo.W1b);
        pk.nat(g + 0, ld0, 2 * H, H, W, d.HT, d.WB, lo.W1c);
        pk.nat(g + 2, H, 0, H, H, d.HT, d.HT, lo.W2);        pk.vec(g + 3, H, d.HP, 1, d.HP, lo.b2);
        pk.matrix(g + 4, 2 * H, 0, H, d.HP, 1, H, d.HP, 2, d.HT, 2 * d.HT, lo.nm0);   pk.vec(g + 5, H, d.HP, 1, d.HP, lo.nm0b);
        pk.nat(g + 6, H, 0, H, H, d.HT, d.HT, lo.nm1);       pk.vec(g + 7, H, d.HP, 1, d.HP, lo.nm1b);
        pk.nat(g + 8, H, 0, W, H, d.WB, d.HT, lo.W3);        pk.vec(g + 9, W, d.WP, 1, d.WP, lo.b3);
        pk.vec(g + 10, H, d.HP, 1, d.HP, lo.watt);           pk.raw(g + 11, 1, lo.batt);
        pk.vec(g + 12, H, d.HP, 1, d.HP, lo.ln_g_w);         pk.vec(g + 13, H, d.HP, 1, d.HP, lo.ln_g_b);
        const int m = pi.msg0 + 9 * l;    // dir_proj.0 w,b | dir_proj.2 w,b | x_proj.0 w | x_proj.2 w | rbf_proj w | ln w,b
        pk.nat(m + 0, W, 0, 3 * H, W, d.D1T, d.WB, lo.dp0);   pk.vec(m + 1, 3 * H, d.D1P, 1, d.D1P, lo.dp0b);
        pk.matrix(m + 2, 3 * H, 0, H, d.HP, 3, 3 * H, d.D1P, 1, 3 * d.HT, d.D1T, lo.dp2);
        pk.vec(m + 3, H, d.HP, 3, 3 * d.HP, lo.dp2b);
        pk.nat(m + 4, H, 0, H, H, d.HT, d.HT, lo.xp0);
        pk.matrix(m + 5, H, 0, H, d.HP, 3, H, d.HP, 1, 3 * d.HT, d.HT, lo.xp2);
        pk.matrix(m + 6, R, 0, H, d.HP, 3, R, d.RP, 1, 3 * d.HT, d.RB, lo.rbfp);
        pk.vec(m + 7, H, d.HP, 1, d.HP, lo.ln_q_w);          pk.vec(m + 8, H, d.HP, 1, d.HP, lo.ln_q_b);
        const int u = pi.upd0 + 9 * l;    // vec_proj w | xvec_proj.0 w | xvec_proj.2 w | lin3.0 w,b | lin3.2 w,b | lin3.4 w,b
        pk.matrix(u + 0, H, 0, H, d.HP, 2, H, d.HP, 1, 2 * d.HT, d.HT, lo.vp);
        pk.matrix(u + 1, 2 * H, 0, H, d.HP, 1, H, d.HP, 2, d.HT, 2 * d.HT, lo.xv0);
        pk.matrix(u + 2, H, 0, H, d.HP, 3, H, d.HP, 1, 3 * d.HT, d.HT, lo.xv2);
        pk.raw(u + 3, 144, lo.l3u); pk.raw(u + 4, 48, lo.l3u + 144); pk.raw(u + 5, 384, lo.l3u + 192);
        pk.raw(u + 6, 8, lo.l3u + 576); pk.raw(u + 7, 8, lo.l3u + 584); pk.raw(u + 8, 1, lo.l3u + 592);
        // ---- LDS weight streams (consumption order, see oard_edge_v1.h) ----
        {
            const size_t G2 = (size_t)(d.HT + 1) * 256;
            const size_t s1 = lo.gcl_stream, s2 = s1 + (size_t)d.WB * d.HT * 256, s3 = s2 + (size_t)(d.HT + 1) * G2;
            // GclStream::TAIL1 / ROWS4 (the same condition): compact K tail of the H-wide inputs of S2 / S3, and the 13th output
            // tile of W1c / W2 and the gate packed for the 4x4x1 MFMA
            const int tc = (H % 16 >= 1 && H % 16 <= 4 && d.HT >= 3 && (d.HT & 1)) ? 1 : 0;
            pk.matrix(g + 0, ld0, 2 * H, H, d.HP, 1, W, d.WP, 1, d.HT, d.WB, s1, 256, (size_t)d.HT * 256, 0, 0, 0, tc * d.HT);   // W1c, K-outer
            pk.matrix(g + 2, H, 0, H, d.HP, 1, H, d.HP, 1, d.HT, d.HT, s2 + 256, G2, 256, 0, 0, tc, tc * d.HT);  // W2 tiles
            pk.bias_chunks(g + 3, H, d.HP, 1, d.HT, s2, G2);
            pk.matrix(g + 10, H, 0, 1, 16, 1, H, d.HP, 1, 1, d.HT, s2 + d.HT * G2 + 256, G2, 256, 0, 0, tc, tc); // watt as a 1-row tile
            pk.bias_chunks(g + 11, 1, 16, 1, 1, s2 + d.HT * G2, G2);
            pk.matrix(g + 8, H, 0, W, d.WP, 1, H, d.HP, 1, d.WB, d.HT, s3 + 256, G2, 256, 0, 0, tc);             // W3 tiles
            pk.bias_chunks(g + 9, W, d.WP, 1, d.WB, s3, G2);
            const size_t GE = (size_t)(1 + d.D1T + d.RB) * 256;
            const size_t t1 = lo.equi_stream, t2 = t1 + (size_t)d.WB * d.D1T * 256;
            pk.matrix(m + 0, W, 0, 3 * H, d.D1P, 1, W, d.WP, 1, d.D1T, d.WB, t1, 256, (size_t)d.D1T * 256);       // dir_proj.0, K-outer
            const int r4 = (H % 16 >= 1 && H % 16 <= 4) ? d.HT : 0;     // EquiStream::ROWS4: the 13th tile of every third
            pk.matrix(m + 2, 3 * H, 0, H, d.HP, 3, 3 * H, d.D1P, 1, 3 * d.HT, d.D1T, t2 + 256, GE, 256, d.HT, 0, 0, r4);    // dir_proj.2 tiles
            pk.matrix(m + 6, R, 0, H, d.HP, 3, R, d.RP, 1, 3 * d.HT, d.RB, t2 + (size_t)(1 + d.D1T) * 256, GE, 256, d.HT, 0, 0, r4);
            pk.bias_chunks(m + 3, H, d.HP, 3, 3 * d.HT, t2, GE, d.HT);
        }
    }
    { int rcf = pk.flush(); if (rcf != OARD_OK) return rcf; }
    {   // EquiUpdate's frame-scalar MLP as a checked table (oard_layout.h: L3T_*), from the raw block packed above
        L3tJobs lj;
        memset(&lj, 0, sizeof(lj));
        for (int l = 0; l < c->num_layers; ++l) { lj.l3u[l] = po.layer[l].l3u; lj.l3t[l] = po.layer[l].l3t; }
        hipLaunchKernelGGL(k_lin3u_table_fill, dim3(L3T_N / 256, (unsigned)c->num_layers), dim3(256), 0, st, (float*)packed, lj);
        hipLaunchKernelGGL(k_lin3u_table_check, dim3(L3T_N / 256, (unsigned)c->num_layers), dim3(256), 0, st, (float*)packed, lj);
    }
    hipLaunchKernelGGL(k_c0row, dim3((unsigned)cdiv(d.WP, 256)), dim3(256), 0, st, params[pi.lin30_b], params[pi.lin32_w],
                       params[pi.lin32_b], params[pi.rl0_b], params[pi.rl2_w], params[pi.rl2_b],
                       (float*)packed + po.c0row, H, d.H4, d.WP);
    hipLaunchKernelGGL(k_u0, dim3((unsigned)cdiv(d.HP, 64)), dim3(64), 0, st, params[pi.gcl0 + 0],
                       (const float*)packed + po.c0row, (float*)packed + po.u0, H, W, d.HP);
    if (c->precision & (OARD_PREC_GCL_BF16X3 | OARD_PREC_TRAIN_BF16X3)) {          // the split-precision stream of the GCL kernel, from the natural fp32 packs made above (oard_edge_b3.h)
        const int nbh = (d.HT + 1) / 2, nbw = (d.WB + 1) / 2, G1 = 3 * d.HT, G2 = 1 + 3 * nbh;
        const size_t G2f = (size_t)(d.HT + 1) * 256;                              // group stride of the fp32 stream's S2 / S3 (bias chunk first)
        for (int l = 0; l < c->num_layers; ++l) {
            const LayerOff& lo = po.layer[l];
            const size_t s1 = lo.gcl_b3, s2 = s1 + (size_t)nbw * G1 * 256, s3 = s2 + (size_t)(d.HT + 1) * G2 * 256;
            const size_t f2 = lo.gcl_stream + (size_t)d.WB * d.HT * 256, f3 = f2 + (size_t)(d.HT + 1) * G2f;
            B3Jobs jb;
            jb.j[0] = B3Job{lo.W1c, d.WB, d.HT, nbw, 0, s1, (size_t)3 * 256, (size_t)G1 * 256};                // S1: K-outer
            jb.j[1] = B3Job{lo.W2, d.HT, d.HT, nbh, 0, s2 + 256, (size_t)G2 * 256, (size_t)3 * 256};           // S2: W2 tiles
            jb.j[2] = B3Job{lo.watt, d.HT, 1, nbh, 1, s2 + (size_t)d.HT * G2 * 256 + 256, (size_t)G2 * 256, (size_t)3 * 256};   // gate row
            jb.j[3] = B3Job{lo.W3, d.HT, d.WB, nbh, 0, s3 + 256, (size_t)G2 * 256, (size_t)3 * 256};           // S3: W3 tiles
            const long long work = std::max<long long>((long long)d.HT * nbw, (long long)d.WB * nbh) * 64;
            hipLaunchKernelGGL(k_pack_b3, dim3((unsigned)cdiv(work, 256), 4), dim3(256), 0, st, jb, (float*)packed);
            hipLaunchKernelGGL(k_copy_chunks, dim3((unsigned)(d.HT + 1)), dim3(256), 0, st, (float*)packed, f2, G2f, s2, (size_t)G2 * 256, d.HT + 1);
            hipLaunchKernelGGL(k_copy_chunks, dim3((unsigned)d.WB), dim3(256), 0, st, (float*)packed, f3, G2f, s3, (size_t)G2 * 256, d.WB);
        }
    }
    if (c->precision & (OARD_PREC_EQUI_BF16X3 | OARD_PREC_TRAIN_BF16X3)) {         // the split-precision stream of the EquiMessage kernel: three K-outer sections
        const int nbw = (d.WB + 1) / 2, nbr = (d.RB + 1) / 2, nbd = (d.D1T + 1) / 2, NO = 3 * d.HT, G1 = 3 * d.D1T, G2 = 3 * NO;
        for (int l = 0; l < c->num_layers; ++l) {
            const LayerOff& lo = po.layer[l];
            const size_t t1 = lo.equi_b3, t2a = t1 + (size_t)nbw * G1 * 256, t2b = t2a + (size_t)nbr * G2 * 256;
            B3Jobs jb;
            jb.j[0] = B3Job{lo.dp0, d.WB, d.D1T, nbw, 0, t1, (size_t)3 * 256, (size_t)G1 * 256};
            jb.j[1] = B3Job{lo.rbfp, d.RB, NO, nbr, 0, t2a, (size_t)3 * 256, (size_t)G2 * 256};
            jb.j[2] = B3Job{lo.dp2, d.D1T, NO, nbd, 0, t2b, (size_t)3 * 256, (size_t)G2 * 256};
            jb.j[3] = B3Job{0, 0, 0, 0, 0, 0, 0, 0};
            const long long work = std::max<long long>((long long)d.D1T * nbw, (long long)NO * nbd) * 64;
            hipLaunchKernelGGL(k_pack_b3, dim3((unsigned)cdiv(work, 256), 3), dim3(256), 0, st, jb, (float*)packed);
        }
    }
    HIP_TRY(hipGetLastError());
    return OARD_OK;
}

// ------------------------------------------------------------------------------------------------
// topology
// ------------------------------------------------------------------------------------------------
void oard_topology_destroy(oard_topology* tp);

// ---- the library's own streams (round 4) --------------------------------------------------------------------------------------------------
// THREE non-blocking streams per device, created once: the sub-batch streams of every multi-part topology (side[1..3]), the gradient
// stream of the training sweep ([0]) and the stream of the table uploads ([2]).  The ROCm runtime serves a process's streams from 4
// hardware queues; with the caller's stream these make exactly 4.  Streams created per topology / per purpose (rounds 1-3) made the queue a
// compute stream lands on depend on what had been created before it: one extra stream and two sub-batches of the B = 64 denoising step
// shared a queue (17.9 -> 20.5 ms).
// Which hardware queue a stream lands on depends on how many streams the process created before it (measured, tools/queue_probe.py: 1, 2
// or 4 foreign streams created first and the denoising step is 20.3 instead of 17.9 ms - RCCL creates streams, so does any host program).
// So the three streams are CHOSEN: up to 8 candidates are created and a candidate is taken if a 300-us spin kernel on it overlaps with
// the same kernel on the null stream and on every stream taken so far (wall time of the pair < 1.6 x the time of one).  ~10 ms, once per
// device; OARD_NO_STREAM_CALIBRATION=1 takes the first three.
__global__ void k_spin(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
static double spin_pair_us(hipStream_t a, hipStream_t b, bool both) {
    const long long ticks = 30000;                        // ~300 us of the 100-MHz wall clock (measured below, not assumed)
    (void)hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, a, ticks);
    if (both) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, b, ticks);
    (void)hipStreamSynchronize(a);
    if (both) (void)hipStreamSynchronize(b);
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}
static bool streams_overlap(hipStream_t a, hipStream_t b, double one_us) { return spin_pair_us(a, b, true) < 1.6 * one_us; }
static hipStream_t* device_streams() {
    static hipStream_t all[64][3] = {};
    static std::mutex m;
    int d = 0;
    (void)hipGetDevice(&d);
    d &= 63;
    std::lock_guard<std::mutex> lk(m);
    if (!all[d][0]) {
        hipStream_t cand[8] = {};
        int n_cand = 0, taken = 0;
        for (; n_cand < 8; ++n_cand)
            if (hipStreamCreateWithFlags(&cand[n_cand], hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); break; }
        const bool calibrate = getenv("OARD_NO_STREAM_CALIBRATION") == nullptr;
        bool used[8] = {};
        double one_us = 0;
        if (calibrate && n_cand > 0) { (void)spin_pair_us(nullptr, nullptr, false); one_us = spin_pair_us(nullptr, nullptr, false); }
        for (int c = 0; c < n_cand && taken < 3 && calibrate; ++c) {
            bool ok = streams_overlap(nullptr, cand[c], one_us);
            for (int k = 0; k < taken && ok; ++k) ok = streams_overlap(all[d][k], cand[c], one_us);
            if (ok) { all[d][taken++] = cand[c]; used[c] = true; }
        }
        for (int c = 0; c < n_cand && taken < 3; ++c)      // not enough independent queues (or calibration off): take what there is
            if (!used[c]) { all[d][taken++] = cand[c]; used[c] = true; }
        for (int c = 0; c < n_cand; ++c)
            if (!used[c]) (void)hipStreamDestroy(cand[c]);
        (void)hipDeviceSynchronize();
        (void)hipGetLastError();
    }
    return all[d];
}

extern "C" int oard_library_stream(int which, oard_stream_t* out) {
    if (!out || which < 0 || which > 2) return OARD_EINVAL;
    *out = (oard_stream_t)device_streams()[which];
    return OARD_OK;
}

// ---- table pool (round 4) ---------------------------------------------------------------------------------------------------------------
// Training sees a new batch layout every step, so a topology is built and dropped per step.  hipMalloc / blocking hipMemcpy on the null
// stream / hipFree each wait for the device - a pipeline drain per step (measured: + 18 ms on a 68-ms step).  Instead: device blocks and
// pinned staging blocks come from free lists (capacity-matched, at most 2 x the request); the tables are packed into ONE pinned block and
// uploaded by ONE hipMemcpyAsync on the library's third stream (device_streams) - the host does not wait, and in training, where that
// stream is otherwise idle, the copy runs beside the previous step - with an event behind it that every stream waits for before its first
// use of the topology; a destroyed topology's blocks return to the lists behind events recorded on the streams that used it.
struct TablePool {
    struct Free { void* p; size_t cap; };
    struct Pending { void* p; size_t cap; hipEvent_t ev[4]; int n_ev; int kind; };
    std::mutex m;
    std::vector<Free> free_[2];                        // kind 0: device memory, 1: pinned host memory
    std::vector<Pending> pending;
    std::vector<hipEvent_t> spare_events;
};
static TablePool& table_pool() {
    static TablePool pools[64];
    int d = 0;
    (void)hipGetDevice(&d);
    return pools[d & 63];
}
static void pool_raw_free(void* p, int kind) { if (kind == 0) (void)hipFree(p); else (void)hipHostFree(p); }
static void pool_collect(TablePool& tp) {               // pending -> free where every event has fired (non-blocking)
    for (size_t i = 0; i < tp.pending.size();) {
        TablePool::Pending& q = tp.pending[i];
        bool done = true;
        for (int k = 0; k < q.n_ev && done; ++k) done = hipEventQuery(q.ev[k]) == hipSuccess;
        if (!done) { ++i; continue; }
        for (int k = 0; k < q.n_ev; ++k) tp.spare_events.push_back(q.ev[k]);
        tp.free_[q.kind].push_back({q.p, q.cap});
        tp.pending[i] = tp.pending.back();
        tp.pending.pop_back();
    }
    (void)hipGetLastError();                              // hipEventQuery's hipErrorNotReady is not an error
}
static void* pool_alloc(size_t bytes, size_t* cap, int kind = 0) {
    TablePool& tp = table_pool();
    std::lock_guard<std::mutex> lk(tp.m);
    pool_collect(tp);
    std::vector<TablePool::Free>& fl = tp.free_[kind];
    int best = -1;
    for (int i = 0; i < (int)fl.size(); ++i)
        if (fl[i].cap >= bytes && fl[i].cap <= 2 * bytes + (1u << 20) && (best < 0 || fl[i].cap < fl[best].cap)) best = i;
    if (best >= 0) {
        void* p = fl[best].p;
        *cap = fl[best].cap;
        fl[best] = fl.back();
        fl.pop_back();
        return p;
    }
    while (fl.size() > 16) {                              // blocks of shapes that no longer occur
        pool_raw_free(fl.front().p, kind);
        fl.erase(fl.begin());
    }
    void* p = nullptr;
    *cap = align_up(bytes + bytes / 8, (size_t)1 << 16);  // some slack: the next layout of about this size fits as well
    const hipError_t e = kind == 0 ? hipMalloc(&p, *cap) : hipHostMalloc(&p, *cap, hipHostMallocDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
// returns the block to the pool; usable again once everything enqueued so far on `streams` (and on the upload stream) has been executed
static void pool_release(void* p, size_t cap, const hipStream_t* streams, int n_streams, bool sync_device, int kind = 0) {
    if (!p) return;
    TablePool& tp = table_pool();
    std::lock_guard<std::mutex> lk(tp.m);
    if (sync_device) { (void)hipDeviceSynchronize(); tp.free_[kind].push_back({p, cap}); return; }
    TablePool::Pending q{p, cap, {}, 0, kind};
    hipStream_t all[5] = {device_streams()[2]};           // the upload stream first, then the streams of use
    int n = 1;
    for (int k = 0; k < n_streams && k < 4; ++k) if (streams[k] != all[0]) all[n++] = streams[k];
    for (int k = 0; k < n && q.n_ev < 4; ++k) {
        hipEvent_t e = nullptr;
        if (!tp.spare_events.empty()) { e = tp.spare_events.back(); tp.spare_events.pop_back(); }
        else if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipDeviceSynchronize(); tp.free_[kind].push_back({p, cap}); return; }
        (void)hipEventRecord(e, all[k]);
        q.ev[q.n_ev++] = e;
    }
    if (n > 4) { (void)hipDeviceSynchronize(); }          // (4 user streams + the upload stream: wait instead of a fifth event)
    tp.pending.push_back(q);
}
// packs `items` into a pinned block and enqueues ONE asynchronous copy on the upload stream; the pinned block (returned) stays with the
// topology until it is destroyed
struct UploadItem { const void* src; size_t bytes; size_t off; };
static int pool_upload(void* dev, const std::vector<UploadItem>& items, size_t total, void** stage_out, size_t* stage_cap) {
    char* stage = (char*)pool_alloc(total, stage_cap, 1);
    if (!stage) return OARD_EHIP;
    for (const UploadItem& it : items) memcpy(stage + it.off, it.src, it.bytes);
    if (hipMemcpyAsync(dev, stage, total, hipMemcpyHostToDevice, device_streams()[2]) != hipSuccess) {
        (void)hipGetLastError();
        pool_release(stage, *stage_cap, nullptr, 0, true, 1);
        return OARD_EHIP;
    }
    *stage_out = stage;
    return OARD_OK;
}
// an entry point is about to enqueue work on `st` that reads the topology's tables: the stream's first use waits for the upload
static void topo_touch(const oard_topology* tp, hipStream_t st) {
    if (!tp) return;
    for (int i = 0; i < tp->n_used; ++i) if (tp->used_on[i] == st) return;
    if (tp->ready) (void)hipStreamWaitEvent(st, tp->ready, 0);
    if (tp->n_used < 4) tp->used_on[tp->n_used++] = st; else tp->used_many = true;
}

// builds the device tables of the sub-batch made of the samples with dense index in [lo, hi)
static int build_part(const oard_config* c, const int64_t* cm, const int64_t* nfs, int N_all, const std::vector<int>& obj_start,
                      const std::vector<int>& dense, const std::vector<long long>& ref_ptr_ref, int lo, int hi,
                      TopoPart& part, int& max_group, int& max_ns) {
    const int n_obj = c->n_obj, B = hi - lo;
    std::vector<int> order;
    for (int i = 0; i < N_all; ++i) if (dense[i] >= lo && dense[i] < hi) order.push_back(i);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        if (dense[a] != dense[b]) return dense[a] < dense[b];
        return nfs[a] < nfs[b];
    });
    const int N = (int)order.size();
    if (N < 1) return OARD_EINVAL;
    std::vector<int> node_obj(N), node_row(N), node_ref(N), node_tidx(N), node_sample(N), sample_ptr(B + 1, 0),
        grp_ptr((size_t)B * n_obj + 1, 0), edge_ptr(N), act_ptr(N + 1, 0);
    for (int n = 0; n < N; ++n) {
        const int r = order[n], sb = dense[r] - lo;
        node_ref[n] = r; node_obj[n] = (int)nfs[r]; node_row[n] = r - obj_start[nfs[r]];
        node_tidx[n] = (int)cm[r]; node_sample[n] = sb;
        sample_ptr[sb + 1]++;
        grp_ptr[(size_t)sb * n_obj + nfs[r] + 1]++;
    }
    for (int b = 0; b < B; ++b) sample_ptr[b + 1] += sample_ptr[b];
    for (size_t q = 0; q < (size_t)B * n_obj; ++q) grp_ptr[q + 1] += grp_ptr[q];
    long long E = 0, A = 0;
    for (int b = 0; b < B; ++b) {
        const long long ns = sample_ptr[b + 1] - sample_ptr[b];
        max_ns = std::max(max_ns, (int)ns);
        for (int n = sample_ptr[b]; n < sample_ptr[b + 1]; ++n) { edge_ptr[n] = (int)E; E += ns - 1; }
        if (E > 0x7fffffffLL) return OARD_EINVAL;
    }
    for (size_t q = 0; q < (size_t)B * n_obj; ++q) {
        const long long ng = grp_ptr[q + 1] - grp_ptr[q];
        max_group = std::max(max_group, (int)ng);
        A += ng * (ng - 1);
    }
    if (max_group > OARD_MAX_GROUP || A > 0x7fffffffLL) return OARD_EINVAL;
    // one spare entry (index E / A) for the padding columns of the edge kernels
    std::vector<int> edge_src((size_t)E + 1, 0), edge_tgt((size_t)E + 1, 0);
    for (int b = 0; b < B; ++b) {
        const int s0 = sample_ptr[b], s1 = sample_ptr[b + 1];
        for (int n = s0; n < s1; ++n) {
            size_t e = (size_t)edge_ptr[n];
            for (int m = s0; m < s1; ++m) if (m != n) { edge_src[e] = n; edge_tgt[e] = m; ++e; }
        }
    }
    std::vector<int> act_src((size_t)A + 1, 0), act_tgt((size_t)A + 1, 0), act_edge((size_t)A + 1, (int)E);
    {
        size_t a = 0;
        for (int n = 0; n < N; ++n) {
            const int q = node_sample[n] * n_obj + node_obj[n];
            const int g0 = grp_ptr[q], g1 = grp_ptr[q + 1], s0 = sample_ptr[node_sample[n]];
            act_ptr[n] = (int)a;
            for (int m = g0; m < g1; ++m) if (m != n) {
                act_src[a] = m; act_tgt[a] = n;
                act_edge[a] = edge_ptr[m] + (n - s0) - (n > m ? 1 : 0);
                ++a;
            }
        }
        act_ptr[N] = (int)a;
    }
    // physical rows: inner edges first (row a == inner entry a, target-sorted), then inter-object edges in
    // logical order, then the spare row
    std::vector<int> edge_row((size_t)std::max<long long>(E, 1), -1), row_src((size_t)E + 1, 0), row_tgt((size_t)E + 1, 0),
        row_eid((size_t)E + 1, (int)E);
    {
        for (long long a = 0; a < A; ++a) {
            edge_row[act_edge[a]] = (int)a; row_src[a] = act_src[a]; row_tgt[a] = act_tgt[a]; row_eid[a] = act_edge[a];
        }
        long long r = A;
        for (long long e = 0; e < E; ++e)
            if (edge_row[e] < 0) { edge_row[e] = (int)r; row_src[r] = edge_src[e]; row_tgt[r] = edge_tgt[e]; row_eid[r] = (int)e; ++r; }
    }
    std::vector<long long> ref_edge_ptr(N);
    for (int n = 0; n < N; ++n) ref_edge_ptr[n] = ref_ptr_ref[node_ref[n]];

    std::vector<UploadItem> items;
    size_t cur = 0;
    auto add = [&](const void* p, size_t bytes) { UploadItem it{p, bytes, cur}; cur = align_up(cur + bytes, 256); items.push_back(it); return it.off; };
    const size_t o_obj = add(node_obj.data(), N * 4), o_row = add(node_row.data(), N * 4), o_ref = add(node_ref.data(), N * 4),
                 o_tidx = add(node_tidx.data(), N * 4), o_smp = add(node_sample.data(), N * 4),
                 o_sptr = add(sample_ptr.data(), (B + 1) * 4), o_eptr = add(edge_ptr.data(), N * 4),
                 o_esrc = add(edge_src.data(), edge_src.size() * 4), o_etgt = add(edge_tgt.data(), edge_tgt.size() * 4),
                 o_gptr = add(grp_ptr.data(), grp_ptr.size() * 4), o_aptr = add(act_ptr.data(), (N + 1) * 4),
                 o_asrc = add(act_src.data(), act_src.size() * 4), o_atgt = add(act_tgt.data(), act_tgt.size() * 4),
                 o_aedge = add(act_edge.data(), act_edge.size() * 4), o_rptr = add(ref_edge_ptr.data(), N * 8),
                 o_erow = add(edge_row.data(), edge_row.size() * 4), o_rsrc = add(row_src.data(), row_src.size() * 4),
                 o_rtgt = add(row_tgt.data(), row_tgt.size() * 4), o_reid = add(row_eid.data(), row_eid.size() * 4);
    size_t cap = 0;
    char* dev = (char*)pool_alloc(cur, &cap);
    if (!dev) return OARD_EHIP;
    if (int rc = pool_upload(dev, items, cur, &part.stage, &part.stage_cap)) { pool_release(dev, cap, nullptr, 0, false); return rc; }
    part.dev_cap = cap;
    part.dev_block = dev;
    TopoDev& d = part.d;
    d.N = N; d.B = B; d.n_obj = n_obj; d.n_groups = B * n_obj; d.E = E; d.A = A;
    d.node_obj = (const int*)(dev + o_obj); d.node_row = (const int*)(dev + o_row); d.node_ref = (const int*)(dev + o_ref);
    d.node_tidx = (const int*)(dev + o_tidx); d.node_sample = (const int*)(dev + o_smp); d.sample_ptr = (const int*)(dev + o_sptr);
    d.edge_ptr = (const int*)(dev + o_eptr); d.edge_src = (const int*)(dev + o_esrc); d.edge_tgt = (const int*)(dev + o_etgt);
    d.grp_ptr = (const int*)(dev + o_gptr); d.act_ptr = (const int*)(dev + o_aptr); d.act_src = (const int*)(dev + o_asrc);
    d.act_tgt = (const int*)(dev + o_atgt); d.act_edge = (const int*)(dev + o_aedge);
    d.ref_edge_ptr = (const long long*)(dev + o_rptr);
    d.edge_row = (const int*)(dev + o_erow); d.row_src = (const int*)(dev + o_rsrc); d.row_tgt = (const int*)(dev + o_rtgt);
    d.row_eid = (const int*)(dev + o_reid);
    return OARD_OK;
}

int oard_topology_create(const oard_config* c, const int64_t* cm, const int64_t* nfs, int64_t n_nodes,
                         oard_topology** out) {
    return oard_topology_create_parts(c, cm, nfs, n_nodes, g_parts, out);
}

int oard_topology_create_parts(const oard_config* c, const int64_t* cm, const int64_t* nfs, int64_t n_nodes,
                               int parts, oard_topology** out) {
    if (!config_ok(c) || !cm || !nfs || !out || n_nodes < 1 || n_nodes > (1 << 24)) return OARD_EINVAL;
    const int N = (int)n_nodes, n_obj = c->n_obj;
    std::vector<int> obj_start(n_obj + 1, 0);
    for (int i = 0; i < N; ++i) {
        if (nfs[i] < 0 || nfs[i] >= n_obj) return OARD_EINVAL;
        if (i > 0 && nfs[i] < nfs[i - 1]) return OARD_EINVAL;           // objects must be contiguous, ascending
        obj_start[nfs[i] + 1]++;
    }
    for (int k = 0; k < n_obj; ++k) obj_start[k + 1] += obj_start[k];
    std::vector<int64_t> samples(cm, cm + N);
    std::sort(samples.begin(), samples.end());
    samples.erase(std::unique(samples.begin(), samples.end()), samples.end());
    const int B = (int)samples.size();
    if (samples.front() < 0 || samples.back() > (1 << 30)) return OARD_EINVAL;
    std::vector<int> dense(N), ns_of(B, 0);
    for (int i = 0; i < N; ++i) {
        dense[i] = (int)(std::lower_bound(samples.begin(), samples.end(), cm[i]) - samples.begin());
        ns_of[dense[i]]++;
    }
    // reference-order edge offsets: prefix of (n_s - 1) over the reference node order (whole batch)
    std::vector<long long> ref_ptr_ref(N);
    long long E_all = 0;
    for (int r = 0; r < N; ++r) { ref_ptr_ref[r] = E_all; E_all += ns_of[dense[r]] - 1; }

    int n_parts = parts > 0 ? parts : (B >= 32 ? 4 : (B >= 16 ? 2 : 1));
    n_parts = std::max(1, std::min(std::min(n_parts, OARD_MAX_PARTS), B));
    oard_topology* tp = new oard_topology();
    (void)hipGetDevice(&tp->device);
    tp->n_obj = n_obj; tp->B = B; tp->n_parts = n_parts;
    {   // reference-order tables of oard_topology_check_edge_index: sample, rank inside the sample, first edge id of every node
        std::vector<int> tab(2 * (size_t)N), seen(B, 0);
        for (int r = 0; r < N; ++r) { tab[r] = dense[r]; tab[(size_t)N + r] = seen[dense[r]]++; }
        const size_t b_int = 2 * (size_t)N * sizeof(int), b_all = align_up(b_int, 8) + (size_t)N * sizeof(long long);
        tp->ref_block = pool_alloc(b_all, &tp->ref_cap);
        if (!tp->ref_block) { delete tp; return OARD_EHIP; }
        const std::vector<UploadItem> ref_items = {{tab.data(), b_int, 0}, {ref_ptr_ref.data(), (size_t)N * sizeof(long long), align_up(b_int, 8)}};
        if (int rc = pool_upload(tp->ref_block, ref_items, b_all, &tp->ref_stage, &tp->ref_stage_cap)) { oard_topology_destroy(tp); return rc; }
        tp->ref_sample = (const int*)tp->ref_block; tp->ref_rank = tp->ref_sample + N;
        tp->ref_ptr = (const long long*)((char*)tp->ref_block + align_up(b_int, 8));
        tp->N_ref = N;
    }
    for (int k = 0; k <= n_obj && k <= OARD_MAX_OBJECTS; ++k) tp->obj_start[k] = obj_start[k];
    for (int p = 0; p < n_parts; ++p) {
        const int lo = (int)((long long)B * p / n_parts), hi = (int)((long long)B * (p + 1) / n_parts);
        int rc = build_part(c, cm, nfs, N, obj_start, dense, ref_ptr_ref, lo, hi, tp->parts[p], tp->max_group, tp->max_ns);
        if (rc != OARD_OK) { oard_topology_destroy(tp); return rc; }
        tp->N += tp->parts[p].d.N; tp->E += tp->parts[p].d.E; tp->A += tp->parts[p].d.A;
        tp->parts[p].conc = n_parts;
    }
    {   // node-stage workgroup shape: about one workgroup per CU over all concurrent sub-batches, at most 16 nodes each
        const int npb = g_npb > 0 ? g_npb : (int)std::max<long long>(1, std::min<long long>(16, cdiv(tp->N, 256)));
        for (int p = 0; p < n_parts; ++p) tp->parts[p].d.npb = npb;
    }
    if (n_parts > 1) {
        HIP_TRY(hipEventCreateWithFlags(&tp->ev_fork, hipEventDisableTiming));
        for (int p = 1; p < n_parts; ++p) {
            tp->side[p] = device_streams()[(p - 1) % 3];     // shared by all topologies of the device (more than 4 parts: a debug setting)
            if (!tp->side[p]) { oard_topology_destroy(tp); return OARD_EHIP; }
            HIP_TRY(hipEventCreateWithFlags(&tp->ev_join[p], hipEventDisableTiming));
        }
    }
    // the uploads sit on the upload stream: every stream waits for this event before its first use of the topology (topo_touch)
    HIP_TRY(hipEventCreateWithFlags(&tp->ready, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(tp->ready, device_streams()[2]));
    *out = tp;
    return OARD_OK;
}

void oard_topology_destroy(oard_topology* tp) {
    if (!tp) return;
    // reached from Python __del__ with whatever device happens to be current: the pool, the streams and the events are per device, so the
    // blocks go back under the device they were created on (a device-A block on device B's free list would be handed out as B's memory)
    int cur = tp->device;
    (void)hipGetDevice(&cur);
    if (cur != tp->device) (void)hipSetDevice(tp->device);
    struct Restore { int cur, dev; ~Restore() { if (cur != dev) (void)hipSetDevice(cur); } } restore_{cur, tp->device};
    // the side streams of a multi-part topology are joined into the caller's stream at the end of every call, so events on the
    // callers' streams cover their work as well
    for (int p = 0; p < OARD_MAX_PARTS; ++p) {
        pool_release(tp->parts[p].dev_block, tp->parts[p].dev_cap, tp->used_on, tp->n_used, tp->used_many);
        pool_release(tp->parts[p].stage, tp->parts[p].stage_cap, nullptr, 0, false, 1);          // read by the upload stream only
        if (tp->ev_join[p]) (void)hipEventDestroy(tp->ev_join[p]);
    }
    if (tp->ev_fork) (void)hipEventDestroy(tp->ev_fork);
    pool_release(tp->ref_block, tp->ref_cap, tp->used_on, tp->n_used, tp->used_many);
    pool_release(tp->ref_stage, tp->ref_stage_cap, nullptr, 0, false, 1);
    if (tp->ready) (void)hipEventDestroy(tp->ready);
    delete tp;
}
int64_t oard_topology_num_nodes(const oard_topology* tp) { return tp ? tp->N : 0; }
int64_t oard_topology_num_edges(const oard_topology* tp) { return tp ? tp->E : 0; }
int64_t oard_topology_num_inner_edges(const oard_topology* tp) { return tp ? tp->A : 0; }
int64_t oard_topology_num_samples(const oard_topology* tp) { return tp ? tp->B : 0; }

// Outputs are per node and every segmented sum runs in the library's own (implicit) edge order, so ANY ordering of the complete
// edge set is the same computation (egnn_dynamics.py:63-72 accepts any edge_index; utils/_graph_tools.py:30-36 builds the complete
// per-sample graph).  Set check: every given pair must be a valid ordered pair (same sample, i != j) and hit its id - first id of
// node i + rank of j among the other nodes of the sample - exactly once in a bitmap; with n_edges == E that is a permutation.
int oard_topology_check_edge_index(const oard_topology* tp, const int64_t* ei, int64_t n_edges, int32_t* ok,
                                   oard_stream_t stream) {
    topo_touch(tp, (hipStream_t)stream);
    if (!tp || !ok) return OARD_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int one = (n_edges == tp->E) ? 1 : 0;
    HIP_TRY(hipMemcpyAsync(ok, &one, sizeof(int), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));      // `one` lives on this stack frame
    if (one && tp->E > 0) {
        if (!ei) return OARD_EINVAL;
        unsigned* bitmap = nullptr;
        const size_t words = (size_t)cdiv(tp->E, 32);
        HIP_TRY(hipMalloc(&bitmap, words * sizeof(unsigned)));
        hipError_t e1 = hipMemsetAsync(bitmap, 0, words * sizeof(unsigned), st);
        if (e1 == hipSuccess) {
            hipLaunchKernelGGL(k_check_edge_set, dim3((unsigned)cdiv(n_edges, 256)), dim3(256), 0, st, tp->ref_sample, tp->ref_rank,
                               tp->ref_ptr, tp->N_ref, (const long long*)ei, (long long)n_edges, bitmap, (int*)ok);
            e1 = hipGetLastError();
        }
        const hipError_t e2 = hipStreamSynchronize(st);
        (void)hipFree(bitmap);
        HIP_TRY(e1);
        HIP_TRY(e2);
    }
    return OARD_OK;
}

// ------------------------------------------------------------------------------------------------
// workspace
// ------------------------------------------------------------------------------------------------
static size_t ws_total(const oard_config* c, const oard_topology* tp) {
    size_t cur = 0;
    for (int p = 0; p < tp->n_parts; ++p) {
        const_cast<TopoPart&>(tp->parts[p]).ws_off = cur;
        cur = align_up(cur + make_ws(c, tp->parts[p].d).total, 4096);
    }
    return cur;
}
size_t oard_workspace_bytes(const oard_config* c, const oard_topology* tp) {
    if (!config_ok(c) || !tp) return 0;
    return ws_total(c, tp);
}

int oard_debug_lin3u_table(const oard_config* c, const void* packed, int layer, float* out, oard_stream_t stream) {
    static_assert(OARD_L3T_FLOATS == L3T_FLOATS, "include/oard.h and csrc/oard_layout.h disagree on the table size");
    if (!config_ok(c) || !packed || !out || layer < 0 || layer >= c->num_layers) return OARD_EINVAL;
    const PackOff po = make_layout(c);
    HIP_TRY(hipMemcpyAsync(out, (const float*)packed + po.layer[layer].l3t, (size_t)L3T_FLOATS * sizeof(float), hipMemcpyDeviceToDevice,
                           (hipStream_t)stream));
    return OARD_OK;
}

int oard_active_inner_edges(const oard_config* c, const oard_topology* topo, const void* ws, size_t ws_bytes, int64_t* n_active,
                            oard_stream_t stream) {
    if (!config_ok(c) || !topo || !ws || !n_active) return OARD_EINVAL;
    if (ws_bytes < ws_total(c, topo)) return OARD_ENOMEM;
    hipStream_t st = (hipStream_t)stream;       // (decided by what the last call WROTE, not by the current debug options)
    long long total = 0;
    for (int p = 0; p < topo->n_parts; ++p) {
        if (topo->parts[p].d.A <= 0) continue;
        int n = 0;
        const WsOff w = make_ws(c, topo->parts[p].d);
        HIP_TRY(hipMemcpyAsync(&n, (const char*)ws + topo->parts[p].ws_off + w.al_n, sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (n < 0 || n > topo->parts[p].d.A) { *n_active = -1; return OARD_OK; }      // no list was built by the last call on this workspace
        total += n;
    }
    *n_active = total;
    return OARD_OK;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
int oard_forward(const oard_config* c, const oard_topology* topo, const void* packed, const float* const* xh,
                 const float* t, int t_is_scalar, const float* cond, float* const* out, void* ws, size_t ws_bytes,
                 int32_t* status, oard_stream_t stream) {
    topo_touch(topo, (hipStream_t)stream);
    if (!config_ok(c) || !topo || !packed || !xh || !out || !ws || !status) return OARD_EINVAL;
    if (c->condition_time && !t) return OARD_EINVAL;
    if (c->condition_nf > 0 && !cond) return OARD_EINVAL;
    if (c->n_obj != topo->n_obj) return OARD_EINVAL;
    if (ws_bytes < ws_total(c, topo)) return OARD_ENOMEM;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(status, 0, sizeof(int), st));
    if (g_poison) HIP_TRY(hipMemsetAsync(ws, 0xFF, ws_total(c, topo), st));
    // sub-batches run concurrently on the topology's side streams (sequentially when timing / debugging)
    const bool concurrent = topo->n_parts > 1 && !g_timing.on && g_stop_after == 0 && !g_sequential;
    if (concurrent) HIP_TRY(hipEventRecord(topo->ev_fork, st));
    for (int p = 0; p < topo->n_parts; ++p) {
        hipStream_t sp = (concurrent && p > 0) ? topo->side[p] : st;
        if (concurrent && p > 0) HIP_TRY(hipStreamWaitEvent(sp, topo->ev_fork, 0));
        int rc = OARD_EINVAL;
        DISPATCH_DIMS(c, rc = forward_impl<D>(c, &topo->parts[p], (const float*)packed, xh, t, t_is_scalar, cond, out,
                                              (char*)ws + topo->parts[p].ws_off, nullptr, (int*)status, sp));
        if (rc != OARD_OK) return rc;
        if (concurrent && p > 0) HIP_TRY(hipEventRecord(topo->ev_join[p], sp));
    }
    if (concurrent)
        for (int p = 1; p < topo->n_parts; ++p) HIP_TRY(hipStreamWaitEvent(st, topo->ev_join[p], 0));
    return OARD_OK;
}

static int sampler_step_impl(const oard_config* c, const oard_topology* topo, int mode, const float* const* z,
                             const float* const* eh, const float* const* noise, const float* const* h0, float a, float b,
                             float cc, const float* coef, int zero_feature_noise, float* const* out, oard_stream_t stream);

int oard_sampler_step(const oard_config* c, const oard_topology* topo, int mode, const float* const* z,
                      const float* const* eh, const float* const* noise, const float* const* h0, float a, float b,
                      float cc, int zero_feature_noise, float* const* out, oard_stream_t stream) {
    topo_touch(topo, (hipStream_t)stream);
    return sampler_step_impl(c, topo, mode, z, eh, noise, h0, a, b, cc, nullptr, zero_feature_noise, out, stream);
}

int oard_sampler_step_dev(const oard_config* c, const oard_topology* topo, int mode, const float* const* z,
                          const float* const* eh, const float* const* noise, const float* const* h0, const float* coef,
                          int zero_feature_noise, float* const* out, oard_stream_t stream) {
    topo_touch(topo, (hipStream_t)stream);
    if (!coef) return OARD_EINVAL;
    return sampler_step_impl(c, topo, mode, z, eh, noise, h0, 0.f, 0.f, 0.f, coef, zero_feature_noise, out, stream);
}

static int sampler_step_impl(const oard_config* c, const oard_topology* topo, int mode, const float* const* z,
                             const float* const* eh, const float* const* noise, const float* const* h0, float a, float b,
                             float cc, const float* coef, int zero_feature_noise, float* const* out, oard_stream_t stream) {
    if (!config_ok(c) || !topo || !noise || !out || mode < 0 || mode > 4) return OARD_EINVAL;
    if (mode != 2 && !z) return OARD_EINVAL;
    if (mode <= 1 && !eh) return OARD_EINVAL;
    if (c->n_obj != topo->n_obj) return OARD_EINVAL;
    SamplerPtrs sp;
    memset(&sp, 0, sizeof(sp));
    for (int k = 0; k < c->n_obj; ++k) {
        sp.z[k] = mode == 2 ? noise[k] : z[k];
        sp.eh[k] = (mode == 2 || !eh) ? noise[k] : eh[k];
        sp.noise[k] = noise[k];
        sp.h0[k] = h0 ? h0[k] : nullptr;
        sp.out[k] = out[k];
        sp.node_nf[k] = c->node_nf[k];
    }
    hipStream_t st = (hipStream_t)stream;
    for (int p = 0; p < topo->n_parts; ++p)
        LAUNCH(F_OTHER, k_sampler_step, cdiv(topo->parts[p].d.N, 128), 128, st, topo->parts[p].d, sp, mode, a, b, cc, coef,
               zero_feature_noise);
    HIP_TRY(hipGetLastError());
    return OARD_OK;
}

// ------------------------------------------------------------------------------------------------
// training: tape, training-mode forward, backward of the edge stages, weight-gradient GEMM
// ------------------------------------------------------------------------------------------------
int oard_topology_export(const oard_topology* topo, int which, int32_t* dst, int64_t capacity, oard_stream_t stream) {
    topo_touch(topo, (hipStream_t)stream);
    if (!topo || !dst || topo->n_parts != 1) return OARD_EINVAL;
    const TopoDev& d = topo->parts[0].d;
    const int* src = nullptr;
    long long n = 0;
    switch (which) {
        case OARD_TOPO_NODE_REF: src = d.node_ref; n = d.N; break;
        case OARD_TOPO_NODE_OBJ: src = d.node_obj; n = d.N; break;
        case OARD_TOPO_NODE_ROW: src = d.node_row; n = d.N; break;
        case OARD_TOPO_NODE_SAMPLE: src = d.node_sample; n = d.N; break;
        case OARD_TOPO_NODE_TIDX: src = d.node_tidx; n = d.N; break;
        case OARD_TOPO_SAMPLE_PTR: src = d.sample_ptr; n = d.B + 1; break;
        case OARD_TOPO_GROUP_PTR: src = d.grp_ptr; n = d.n_groups + 1; break;
        case OARD_TOPO_INNER_SRC: src = d.act_src; n = d.A; break;
        case OARD_TOPO_INNER_TGT: src = d.act_tgt; n = d.A; break;
        case OARD_TOPO_ROW_SRC: src = d.row_src; n = d.E; break;
        case OARD_TOPO_ROW_TGT: src = d.row_tgt; n = d.E; break;
        default: return OARD_EINVAL;
    }
    if (capacity < n) return OARD_ENOMEM;
    if (n > 0) HIP_TRY(hipMemcpyAsync(dst, src, (size_t)n * sizeof(int), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return OARD_OK;
}

size_t oard_tape_bytes(const oard_config* c, const oard_topology* topo) {
    if (!config_ok(c) || !topo || topo->n_parts != 1) return 0;
    return make_tape(c, topo->parts[0].d).total;
}

int oard_tape_entry(const oard_config* c, const oard_topology* topo, int which, int layer, size_t* offset_bytes,
                    int64_t* rows, int64_t* row_floats) {
    if (!config_ok(c) || !topo || topo->n_parts != 1 || !offset_bytes || !rows || !row_floats) return OARD_EINVAL;
    const TopoDev& d = topo->parts[0].d;
    const TapeOff t = make_tape(c, d);
    const RDims r(c->hidden, c->num_radial);
    const int L = c->num_layers;
    const bool per_layer = which >= OARD_TAPE_S_IN;
    const bool plus1 = which == OARD_TAPE_S_IN || which == OARD_TAPE_VEC_IN || which == OARD_TAPE_EW;
    if (per_layer ? (layer < 0 || layer >= L + (plus1 ? 1 : 0)) : layer != 0) return OARD_EINVAL;
    const int64_t N = d.N, E1 = d.E + 1, A1 = d.A + 1;
    switch (which) {
        case OARD_TAPE_HIN: *offset_bytes = t.hin; *rows = N; *row_floats = 16; break;
        case OARD_TAPE_GEO: *offset_bytes = t.geo; *rows = A1; *row_floats = GEO_STRIDE; break;
        case OARD_TAPE_RBF: *offset_bytes = t.rbuf; *rows = A1; *row_floats = r.RP; break;
        case OARD_TAPE_PP0: *offset_bytes = t.pp0; *rows = N; *row_floats = 1; break;
        case OARD_TAPE_X1: *offset_bytes = t.x1; *rows = N; *row_floats = 3; break;
        case OARD_TAPE_S_IN: *offset_bytes = t.s_in[layer]; *rows = N; *row_floats = r.HP; break;
        case OARD_TAPE_VEC_IN: *offset_bytes = t.vec_in[layer]; *rows = N; *row_floats = 3 * r.HP; break;
        case OARD_TAPE_AGG: *offset_bytes = t.agg[layer]; *rows = N; *row_floats = r.HP; break;
        case OARD_TAPE_S_MID: *offset_bytes = t.s_mid[layer]; *rows = N; *row_floats = r.HP; break;
        case OARD_TAPE_EW: *offset_bytes = t.ew[layer]; *rows = E1; *row_floats = r.WP; break;
        case OARD_TAPE_Z1: *offset_bytes = t.z1[layer]; *rows = E1; *row_floats = r.HP; break;
        case OARD_TAPE_Z2: *offset_bytes = t.z2[layer]; *rows = E1; *row_floats = r.HP; break;
        case OARD_TAPE_ATT: *offset_bytes = t.att[layer]; *rows = E1; *row_floats = 1; break;
        case OARD_TAPE_Z3: *offset_bytes = t.z3[layer]; *rows = E1; *row_floats = r.WP; break;
        case OARD_TAPE_ZD1: *offset_bytes = t.zd1[layer]; *rows = A1; *row_floats = r.D1P; break;
        case OARD_TAPE_CD: *offset_bytes = t.cd[layer]; *rows = A1; *row_floats = 3 * r.HP; break;
        case OARD_TAPE_S_A: *offset_bytes = t.s_a[layer]; *rows = N; *row_floats = r.HP; break;
        case OARD_TAPE_VEC_A: *offset_bytes = t.vec_a[layer]; *rows = N; *row_floats = 3 * r.HP; break;
        default: return OARD_EINVAL;
    }
    return OARD_OK;
}

int oard_forward_train(const oard_config* c, const oard_topology* topo, const void* packed, const float* const* xh,
                       const float* t, int t_is_scalar, const float* cond, float* const* out, void* ws, size_t ws_bytes,
                       void* tape, size_t tape_bytes, int32_t* status, oard_stream_t stream) {
    topo_touch(topo, (hipStream_t)stream);
    if (!config_ok(c) || !topo || !packed || !xh || !out || !ws || !tape || !status) return OARD_EINVAL;
    if (topo->n_parts != 1 || c->n_obj != topo->n_obj) return OARD_EINVAL;
    if (c->condition_time && !t) return OARD_EINVAL;
    if (c->condition_nf > 0 && !cond) return OARD_EINVAL;
    if (ws_bytes < ws_total(c, topo) || tape_bytes < make_tape(c, topo->parts[0].d).total) return OARD_ENOMEM;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(status, 0, sizeof(int), st));
    if (g_poison) {
        HIP_TRY(hipMemsetAsync(ws, 0xFF, ws_total(c, topo), st));
        HIP_TRY(hipMemsetAsync(tape, 0xFF, make_tape(c, topo->parts[0].d).total, st));
    }
    int rc = OARD_EINVAL;
    DISPATCH_DIMS(c, rc = forward_impl<D>(c, &topo->parts[0], (const float*)packed, xh, t, t_is_scalar, cond, out,
                                          (char*)ws + topo->parts[0].ws_off, (char*)tape, (int*)status, st));
    return rc;
}

}  // extern "C"
#include "oard_train_layout.h"
extern "C" {

size_t oard_packed_bwd_bytes(const oard_config* c) {
    return config_ok(c) ? make_node_bwd_layout(c, make_bwd_layout(c).total).total * sizeof(float) : 0;
}

int oard_pack_weights_bwd(const oard_config* c, const float* const* params, size_t n_params, void* packed,
                          size_t packed_bytes, oard_stream_t stream) {
    if (!config_ok(c) || !params || !packed) return OARD_EINVAL;
    const ParamIdx pi(c);
    if (n_params != (size_t)pi.count) return OARD_EINVAL;
    const BwdOff bo = make_bwd_layout(c);
    const NodeBwdOff nbo = make_node_bwd_layout(c, bo.total);       // the node-side transposes follow the edge-kernel streams
    if (packed_bytes < nbo.total * sizeof(float)) return OARD_ENOMEM;
    const RDims d(c->hidden, c->num_radial);
    const int H = d.H, W = d.W;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(packed, 0, nbo.total * sizeof(float), st));
    Packer pk{params, (float*)packed, st, {}, 0};
    pack_node_bwd(c, pk, nbo);
    for (int l = 0; l < c->num_layers; ++l) {
        const int g = pi.gcl0 + 14 * l, m = pi.msg0 + 9 * l;
        const size_t t3 = bo.layer[l].gcl, t2 = t3 + (size_t)d.WB * d.HT * 256, t1 = t2 + (size_t)d.HT * d.HT * 256;
        // W3^T, K-outer over the WB blocks of dz3: chunk (t, b) = W3[16b..][16t..]^T
        // GclBwdStream::TAIL1 / ROWS4 (the forward's condition): compact K tail of the H-wide inputs of T2 / T1, 13th output tile of
        // W3^T / W2^T packed for the 4x4x1 MFMA
        const int tc = (H % 16 >= 1 && H % 16 <= 4 && d.HT >= 3 && (d.HT & 1)) ? 1 : 0;
        pk.matrix(g + 8, H, 0, H, d.HP, 1, W, d.WP, 1, d.HT, d.WB, t3, 256, (size_t)d.HT * 256, 0, 1, 0, tc * d.HT);
        // W2^T tiles
        pk.matrix(g + 2, H, 0, H, d.HP, 1, H, d.HP, 1, d.HT, d.HT, t2, 0, 256, 0, 1, tc, tc * d.HT);
        // W1c^T tiles (edge_mlp.0 columns 2H..2H+W)
        pk.matrix(g + 0, 2 * H + W, 2 * H, W, d.WP, 1, H, d.HP, 1, d.WB, d.HT, t1, 0, 256, 0, 1, tc, 0);
        pk.vec(g + 10, H, d.HP, 1, d.HP, bo.layer[l].watt);
        const size_t u2 = bo.layer[l].equi, u1 = u2 + (size_t)3 * d.HT * d.D1T * 256;
        // dir_proj.2^T, K-outer over the 3*HT blocks of dcd (thirds padded 196 -> 208): rows = dir_proj hidden
        pk.matrix(m + 2, 3 * H, 0, 3 * H, d.D1P, 1, H, d.HP, 3, d.D1T, 3 * d.HT, u2, 256, (size_t)d.D1T * 256, 0, 1);
        // dir_proj.0^T tiles
        pk.matrix(m + 0, W, 0, W, d.WP, 1, 3 * H, d.D1P, 1, d.WB, d.D1T, u1, 0, 256, 0, 1);
    }
    { int rcf = pk.flush(); if (rcf != OARD_OK) return rcf; }
    HIP_TRY(hipGetLastError());
    return OARD_OK;
}

int oard_gcl_backward_dx(const oard_config* c, const oard_topology* topo, const void* packed_bwd, int layer,
                         const void* tape, const float* dagg, float* dew, float* dz3, float* mout, float* dz2,
                         float* da, float* dz1, oard_stream_t stream) {
    topo_touch(topo, (hipStream_t)stream);
    if (!config_ok(c) || !topo || topo->n_parts != 1 || !packed_bwd || !tape || !dagg || !dew || !dz3 || !mout || !dz2 ||
        !da || !dz1 || layer < 0 || layer >= c->num_layers)
        return OARD_EINVAL;
    const TopoDev& tp = topo->parts[0].d;
    if (tp.E == 0) return OARD_OK;
    const TapeOff to = make_tape(c, tp);
    const BwdOff bo = make_bwd_layout(c);
    int rc = OARD_EINVAL;
    DISPATCH_DIMS(c, rc = gcl_backward_impl<D>(c, tp, (const float*)packed_bwd, bo.layer[layer], layer, (const char*)tape, to,
                                               dagg, dew, dz3, mout, dz2, da, dz1, (hipStream_t)stream));
    return rc;
}

int oard_equi_backward_dx(const oard_config* c, const oard_topology* topo, const void* packed_bwd, int layer,
                          const void* tape, const float* dcd, float* dew, float* dzd1, oard_stream_t stream) {
    topo_touch(topo, (hipStream_t)stream);
    if (!config_ok(c) || !topo || topo->n_parts != 1 || !packed_bwd || !tape || !dcd || !dew || !dzd1 || layer < 0 ||
        layer >= c->num_layers)
        return OARD_EINVAL;
    const TopoDev& tp = topo->parts[0].d;
    if (tp.A == 0) return OARD_OK;
    const TapeOff to = make_tape(c, tp);
    const BwdOff bo = make_bwd_layout(c);
    int rc = OARD_EINVAL;
    DISPATCH_DIMS(c, rc = equi_backward_impl<D>(tp, (const float*)packed_bwd + bo.layer[layer].equi, dcd,
                                                (const float*)((const char*)tape + to.zd1[layer]), dew, dzd1, (hipStream_t)stream));
    return rc;
}

int oard_edge_node_sums(const oard_config* c, const oard_topology* topo, const float* dz1, float* dP, float* dQ,
                        oard_stream_t stream) {
    topo_touch(topo, (hipStream_t)stream);
    if (!config_ok(c) || !topo || topo->n_parts != 1 || !dz1 || !dP || !dQ) return OARD_EINVAL;
    const TopoDev& tp = topo->parts[0].d;
    const RDims d(c->hidden, c->num_radial);
    if (d.HP > 256) return OARD_EINVAL;
    LAUNCH(F_GCL_BWD, k_edge_node_sums, tp.N, 256, (hipStream_t)stream, tp, dz1, d.HP, dP, dQ);
    HIP_TRY(hipGetLastError());
    return OARD_OK;
}

int oard_scalarize_backward(const oard_config* c, const oard_topology* topo, const void* packed, const void* tape,
                            const float* ne1, int ld, const float* dew, float* dne1, float* part, oard_stream_t stream) {
    topo_touch(topo, (hipStream_t)stream);
    if (!config_ok(c) || !topo || topo->n_parts != 1 || !packed || !tape || !ne1 || !dew || !dne1 || !part || ld < c->hidden)
        return OARD_EINVAL;
    const TopoDev& tp = topo->parts[0].d;
    const TapeOff to = make_tape(c, tp);
    int rc = OARD_EINVAL;
    DISPATCH_DIMS(c, rc = scalarize_backward_impl<D>(c, tp, (const float*)packed, (const char*)tape, to, ne1, ld, dew, dne1, part,
                                                     (hipStream_t)stream));
    return rc;
}

int oard_equi_msg_backward(const oard_config* c, const oard_topology* topo, const void* tape, int layer, const float* xq,
                           const float* cr, const float* gx, const float* gv, float* dcd, float* dcr, float* dxq, float* dvec,
                           oard_stream_t stream) {
    topo_touch(topo, (hipStream_t)stream);
    if (!config_ok(c) || !topo || topo->n_parts != 1 || !tape || !xq || !cr || !gx || !gv || !dcd || !dcr || !dxq || !dvec ||
        layer < 0 || layer >= c->num_layers || c->hidden > 256)
        return OARD_EINVAL;
    const TopoDev& tp = topo->parts[0].d;
    const TapeOff to = make_tape(c, tp);
    int rc = OARD_EINVAL;
    DISPATCH_DIMS(c, rc = equi_msg_backward_impl<D>(c, tp, (const char*)tape, to, layer, xq, cr, gx, gv, dcd, dcr, dxq, dvec, (hipStream_t)stream));
    return rc;
}

int oard_lin3u_forward(const oard_config* c, const void* packed, int layer, const float* x, int64_t n, float* out,
                       oard_stream_t stream) {
    if (!config_ok(c) || !packed || !x || !out || n < 0 || layer < 0 || layer >= c->num_layers) return OARD_EINVAL;
    const PackOff po = make_layout(c);
    if (n > 0) LAUNCH(F_NODE, k_lin3u_fwd, cdiv(n, 256), 256, (hipStream_t)stream, (const float*)packed + po.layer[layer].l3u, x, (long long)n, out);
    HIP_TRY(hipGetLastError());
    return OARD_OK;
}

int oard_lin3u_backward(const oard_config* c, const void* packed, int layer, const float* x, const float* dout, int64_t n,
                        float* dx, float* xa, float* h1, float* dz1, float* h2a, float* dz2, oard_stream_t stream) {
    if (!config_ok(c) || !packed || !x || !dout || !dx || !xa || !h1 || !dz1 || !h2a || !dz2 || n < 0 || layer < 0 ||
        layer >= c->num_layers)
        return OARD_EINVAL;
    const PackOff po = make_layout(c);
    if (n > 0) LAUNCH(F_NODE, k_lin3u_bwd, cdiv(n, 256), 256, (hipStream_t)stream, (const float*)packed + po.layer[layer].l3u, x, dout,
                      (long long)n, dx, xa, h1, dz1, h2a, dz2);
    HIP_TRY(hipGetLastError());
    return OARD_OK;
}

// ---- plan of the 16 x 16-tile kernel (oard_wgrad_t16.h): a pure function of the shape ------------------------------------------------
struct WgtPlan { int ok, transposed, MT, NT, nPT, nQT, TM, TN, n_chunks, grid; long long rpc; };
// kernel instantiations: (largest wave tile TM x TN, SiLU on the Q operand)
#ifndef OARD_WGT_INSTANCES
#define OARD_WGT_INSTANCES X(6, 7, false) X(4, 7, true) X(4, 7, false) X(5, 7, true) X(5, 3, false) X(5, 5, true) X(6, 5, false)
#endif
static bool wgt_instance(int TM, int TN, bool silu) {
#define X(tm_, tn_, s_) if (TM == tm_ && TN == tn_ && silu == s_) return true;
    OARD_WGT_INSTANCES
#undef X
    return false;
}
static WgtPlan wgt_plan(int ncY, int ncX, long long rows, int x_silu) {
    WgtPlan best;
    memset(&best, 0, sizeof(best));
    if (g_wgrad_t16 <= 0 || rows < 16384 || (ncY & 15) || (ncX & 15)) return best;
    const int tY = ncY / 16, tX = ncX / 16;
    const int transposed = (!x_silu && tX > tY) ? 1 : 0;      // P (dealt 4 ways, up to 22 tiles per workgroup) = the wider operand; SiLU exists for Q only
    const int MT = transposed ? tX : tY, NT = transposed ? tY : tX;
    double best_cost = 0;
    for (int nPT = (int)cdiv(MT, WGT_PT); nPT <= (int)cdiv(MT, WGT_PT) + 1; ++nPT)
        for (int nQT = (int)cdiv(NT, WGT_QT); nQT <= (int)cdiv(NT, WGT_QT) + 2; ++nQT) {
            const int lp = (int)cdiv(MT, nPT), lq = (int)cdiv(NT, nQT);
            const int TM = (int)cdiv(lp, 4), TN = (int)cdiv(lq, 2);
            if (lp < 4 || lq < 2 || !wgt_instance(TM, TN, x_silu != 0)) continue;
            const int ntile = nPT * nQT;
            long long nch = std::max<long long>(8, (long long)g_wgrad_t16 / ntile / 8 * 8);
            nch = std::min(nch, std::max<long long>(1, cdiv(rows, 4 * WGT_R)));
            const long long rpc = align_up((size_t)cdiv(rows, nch), WGT_R);
            nch = cdiv(rows, rpc);
            const long long grid = cdiv(nch, 8) * 8 * ntile;
            // MFMAs per 4 rows of the busiest SIMD (waves w and w + 4: P part w x Q part 0, P part (w + 2) % 4 x Q part 1)
            int simd = 0;
            for (int w = 0; w < 4; ++w)
                simd = std::max(simd, wgt_size(lp, 4, w) * wgt_size(lq, 2, 0) + wgt_size(lp, 4, (w + 2) & 3) * wgt_size(lq, 2, 1));
            const double cost = (double)cdiv(grid, 256) * (double)rpc * simd;
            if (!best.ok || cost < best_cost) {
                best_cost = cost;
                best = WgtPlan{1, transposed, MT, NT, nPT, nQT, TM, TN, (int)nch, (int)grid, rpc};
            }
        }
    return best;
}
// a padded column of X that no logical input feature maps to (the ones column that turns the product's padded output into db), or -1
static int wgt_ones_col(int ncX, int i_len, int i_pad, int MI) {
    if (i_pad > i_len) return i_len;                               // the pad of the first section
    return MI < ncX ? MI : -1;                                     // i_pad == i_len: unsectioned, the first column behind the features
}

size_t oard_wgrad_scratch_bytes(int ncY, int ncX, int64_t rows) {
    if (ncY < 4 || ncX < 4 || rows < 0) return 0;
    const WgradPlan p = wgrad_plan(ncY, ncX, rows);
    size_t big = ((size_t)p.n_chunks * p.PP * p.QP + (size_t)p.n_chunks * std::max(p.PP, p.QP)) * sizeof(float);
    for (int silu = 0; silu < 2; ++silu) {
        const WgtPlan t = wgt_plan(ncY, ncX, rows, silu);
        if (t.ok) big = std::max(big, (size_t)t.n_chunks * t.MT * t.NT * 256 * sizeof(float));
    }
    return std::max(big, (size_t)1024 * 1024 * sizeof(float));        // the small-output path: <= 1024 chunks x <= 1024 outputs
}

// ---- queue of short weight-gradient products (grouped launches, oard_edge_bwd.h: k_wgrad_q / k_wgrad_reduce_q) --------------------------
// Active while a sweep entry point runs with a queue installed (oard_train_stages.h): wgrad_impl appends the per-wave-tile products
// (k_wgrad<false, 7>) instead of launching them; wgq_flush launches the whole table.  Partials live in the queue's own scratch
// region, one slice per job.  Job tables are uploaded per flush through a ring of pinned / device slots (wgq_flush).
struct WgQueue {
    std::vector<WgqJob> jobs;
    char* region = nullptr; size_t region_bytes = 0, used = 0;
    unsigned blocks = 0, rblocks = 0;
    hipStream_t st = nullptr;
};
static thread_local WgQueue* t_wgq = nullptr;
struct WgQueueScope {           // installs a queue for the duration of one sweep entry point (which flushes it before it returns)
    WgQueue q;
    WgQueueScope(char* region, size_t bytes, hipStream_t st) { q.region = region; q.region_bytes = bytes; q.st = st; t_wgq = &q; }
    ~WgQueueScope() { t_wgq = nullptr; }
};
int g_gate_fold = 1;         // 0: att_mlp gradient sums by column-sum passes over [E][H] instead of inside k_gcl_edge_bwd (A/B)
int g_wgrad_queue = 1;       // 0: every product is launched on its own (A/B, bit-identical)
// The job table of a flush travels like the topology tables do (round 5; it used to be cached on the device by content, and a real run
// with ragged molecule sizes - a different `rows` in every job, every step - missed that cache on every flush: hipMalloc + a blocking
// null-stream hipMemcpy per flush, a device-wide synchronisation every 256 flushes): a per-device ring of WGQ_SLOTS slots, each a pinned
// host block + a device block for WGQ_MAX_JOBS jobs, allocated once; a flush fills the next slot's pinned block, enqueues ONE
// hipMemcpyAsync on the queue's own stream in front of its two launches and records an event behind them.  A slot is reused only after
// its event has fired - the host waits there only if WGQ_SLOTS flushes are still in flight (a training step has ~16).
#define WGQ_SLOTS 64
#define WGQ_MAX_JOBS 64
struct WgqRing {
    std::mutex mu;
    WgqJob* host = nullptr;      // [WGQ_SLOTS][WGQ_MAX_JOBS] pinned
    WgqJob* dev = nullptr;       // [WGQ_SLOTS][WGQ_MAX_JOBS]
    hipEvent_t ev[WGQ_SLOTS] = {};
    bool busy[WGQ_SLOTS] = {};
    int next = 0;
};
static WgqRing& wgq_ring() {
    static WgqRing rings[64];
    int d = 0;
    (void)hipGetDevice(&d);
    return rings[d & 63];
}
static int wgq_flush() {
    WgQueue* q = t_wgq;
    if (!q || q->jobs.empty()) return OARD_OK;
    if (q->jobs.size() > WGQ_MAX_JOBS) return OARD_EINVAL;
    WgqRing& rg = wgq_ring();
    const WgqJob* table = nullptr;
    int slot = 0;
    {
        std::lock_guard<std::mutex> lk(rg.mu);
        if (!rg.host) {                          // allocate into locals, publish only when everything exists (a half-built ring would
            const size_t bytes = (size_t)WGQ_SLOTS * WGQ_MAX_JOBS * sizeof(WgqJob);        // be taken for a complete one by the next flush)
            WgqJob *h0 = nullptr, *d0 = nullptr;
            hipEvent_t ev0[WGQ_SLOTS] = {};
            int made = 0;
            bool ok = hipHostMalloc((void**)&h0, bytes, hipHostMallocDefault) == hipSuccess && hipMalloc((void**)&d0, bytes) == hipSuccess;
            for (; ok && made < WGQ_SLOTS; ++made) ok = hipEventCreateWithFlags(&ev0[made], hipEventDisableTiming) == hipSuccess;
            if (!ok) {
                for (int i = 0; i < made; ++i) if (ev0[i]) (void)hipEventDestroy(ev0[i]);
                if (d0) (void)hipFree(d0);
                if (h0) (void)hipHostFree(h0);
                return OARD_EHIP;
            }
            for (int i = 0; i < WGQ_SLOTS; ++i) rg.ev[i] = ev0[i];
            rg.dev = d0;
            rg.host = h0;
        }
        slot = rg.next;
        rg.next = (rg.next + 1) % WGQ_SLOTS;
        if (rg.busy[slot]) HIP_TRY(hipEventSynchronize(rg.ev[slot]));      // only when WGQ_SLOTS flushes are still in flight
        rg.busy[slot] = true;
        WgqJob* h = rg.host + (size_t)slot * WGQ_MAX_JOBS;
        WgqJob* dv = rg.dev + (size_t)slot * WGQ_MAX_JOBS;
        memcpy(h, q->jobs.data(), q->jobs.size() * sizeof(WgqJob));
        HIP_TRY(hipMemcpyAsync(dv, h, q->jobs.size() * sizeof(WgqJob), hipMemcpyHostToDevice, q->st));
        table = dv;
    }
    {
        ScopedLaunch sl_(F_WGRAD, q->st);
        hipLaunchKernelGGL((k_wgrad_q<7>), dim3(q->blocks), dim3(256), 0, q->st, table, (int)q->jobs.size());
        hipLaunchKernelGGL(k_wgrad_reduce_q, dim3(q->rblocks), dim3(256), 0, q->st, table, (int)q->jobs.size());
    }
    {
        std::lock_guard<std::mutex> lk(rg.mu);
        HIP_TRY(hipEventRecord(rg.ev[slot], q->st));            // the slot's pinned and device blocks are free again behind the two launches
    }
    q->jobs.clear(); q->used = 0; q->blocks = q->rblocks = 0;
    HIP_TRY(hipGetLastError());
    return OARD_OK;
}

// dW (row stride ldW: the destination may be a column slice of a wider nn.Linear weight; nullptr: bias only) and db; acc != 0:
// the result is ADDED to the destination (parameters shared between layers, gradient accumulation into .grad)
static int wgrad_impl(const float* dY, int ldY, int ncY, int o_len, int o_pad, int MO, const float* X, int ldX, int ncX,
                      int x_silu, int i_len, int i_pad, int MI, int64_t rows, float* dW, int ldW, float* db, int acc, void* scratch,
                      size_t scratch_bytes, hipStream_t st) {
    if (!dY || !X || (!dW && !db) || !scratch || rows < 0 || (ncY & 3) || (ncX & 3) || (ldY & 3) || (ldX & 3) || ncY > ldY ||
        ncX > ldX || o_len < 1 || i_len < 1 || o_pad < o_len || i_pad < i_len || MO < 1 || MI < 1 || ldW < MI)
        return OARD_EINVAL;
    if (((MO - 1) / o_len) * o_pad + (MO - 1) % o_len >= ncY || ((MI - 1) / i_len) * i_pad + (MI - 1) % i_len >= ncX)
        return OARD_EINVAL;
    if (scratch_bytes < oard_wgrad_scratch_bytes(ncY, ncX, rows)) return OARD_ENOMEM;
    float* partial = (float*)scratch;
    const bool small_out = o_len >= MO && i_len >= MI && MO <= 64 && MI < 64 && MO * (MI + 1) <= 1024;
    const int ones = db ? wgt_ones_col(ncX, i_len, i_pad, MI) : -1;
    const WgtPlan t = (small_out || (db && ones < 0)) ? WgtPlan{} : wgt_plan(ncY, ncX, rows, x_silu);
    if (t.ok) {                                              // long contraction, 16-aligned operands: the 16 x 16-tile kernel
        WgtArgs a;
        a.P = t.transposed ? X : dY; a.Q = t.transposed ? dY : X;
        a.ldP = t.transposed ? ldX : ldY; a.ldQ = t.transposed ? ldY : ldX;
        a.MT = t.MT; a.NT = t.NT; a.nPT = t.nPT; a.nQT = t.nQT; a.r0 = 0; a.r1 = rows; a.rpc = t.rpc; a.n_chunks = t.n_chunks;
        a.ones_side = db ? (t.transposed ? 1 : 2) : 0; a.ones_col = db ? ones : 0;
        a.partial = partial;
        bool launched = false;
#define X(tm_, tn_, s_) if (!launched && t.TM == tm_ && t.TN == tn_ && (x_silu != 0) == s_) { \
            LAUNCH_LDS(F_WGRAD, (k_wgrad_t16<tm_, tn_, s_>), t.grid, 512, WGT_LDS_BYTES, st, a); launched = true; }
        OARD_WGT_INSTANCES
#undef X
        if (!launched) return OARD_EINVAL;
        {   // k_wgrad_t16 leaves TILE-MAJOR partials (1-KiB register images): k_wgt_reduce is their only reducer (oard_wgrad_t16.h)
            ScopedLaunch sl_(F_WGRAD, st);
            hipLaunchKernelGGL(k_wgt_reduce, dim3((unsigned)(t.MT * t.NT)), dim3(256), 0, st, partial, t.n_chunks, t.MT, t.NT, t.transposed,
                               o_len, o_pad, MO, i_len, i_pad, MI, dW, ldW, db, db ? ones : -1, acc);
        }
        HIP_TRY(hipGetLastError());
        return OARD_OK;
    }
    const WgradPlan p = wgrad_plan(ncY, ncX, rows);
    if (x_silu && p.transposed) return OARD_EINVAL;          // SiLU-on-load exists for the narrow operand only (never needed otherwise)
    if (small_out) {      // small, unsectioned outputs
        const int n_chunks = (int)std::max<long long>(1, std::min<long long>(1024, cdiv(rows, 256)));
        const long long rpc = align_up((size_t)cdiv(std::max<long long>(rows, 1), n_chunks), 64);
        const int nch = (int)cdiv(std::max<long long>(rows, 1), rpc);
        ScopedLaunch sl_(F_WGRAD, st);
        hipLaunchKernelGGL(k_wgrad_small, dim3(nch), dim3(256), 0, st, dY, ldY, MO, X, ldX, MI, x_silu, (long long)rows, rpc, partial);
        hipLaunchKernelGGL(k_wgrad_small_reduce, dim3((unsigned)cdiv(MO * (MI + 1), 4)), dim3(256), 0, st, partial, nch, MO, MI, dW, ldW, db, acc);
        HIP_TRY(hipGetLastError());
        return OARD_OK;
    }
    float* bpartial = partial + (size_t)p.n_chunks * p.PP * p.QP;
    const float* Pm = p.transposed ? X : dY;
    const float* Qm = p.transposed ? dY : X;
    const int ldP = p.transposed ? ldX : ldY, ncP = p.transposed ? ncX : ncY, ldQ = p.transposed ? ldY : ldX, ncQ = p.transposed ? ncY : ncX;
    float* psum = (db && !p.transposed) ? bpartial : nullptr;
    float* qsum = (db && p.transposed) ? bpartial : nullptr;
    if (p.lds) {
#define WGL_LAUNCH(SILU_, NT_, PBW_, QGW_) do { \
        constexpr size_t lds_ = (size_t)(2 * WGL_ROWS * 64 * PBW_ + 2 * WGL_ROWS * (16 * NT_ * QGW_ + 16)) * sizeof(float); \
        LAUNCH_LDS(F_WGRAD, (k_wgrad_lds<SILU_, NT_, PBW_, QGW_>), p.n_chunks * p.gy, 512, lds_, st, Pm, ldP, ncP, Qm, ldQ, ncQ, 0LL, \
                   (long long)rows, p.rpc, p.nPB, p.nQG, partial, psum, qsum); } while (0)
#define WGL_SHAPE(NT_, PBW_, QGW_) do { if (x_silu) WGL_LAUNCH(true, NT_, PBW_, QGW_); else WGL_LAUNCH(false, NT_, PBW_, QGW_); } while (0)
        if (p.NT == 7) WGL_SHAPE(7, 4, 2);
        else WGL_SHAPE(8, 4, 2);
#undef WGL_SHAPE
#undef WGL_LAUNCH
    } else {
        WgQueue* q = t_wgq;
        const size_t need = align_up(((size_t)p.n_chunks * p.PP * p.QP + (size_t)p.n_chunks * std::max(p.PP, p.QP)) * sizeof(float), 256);
        if (q && g_wgrad_queue && q->st == st && !x_silu && p.NT == 7 && p.n_chunks * p.gy > 0 && need <= q->region_bytes) {
            // queued: the layer ends with one grouped launch (wgq_flush)
            if (q->used + need > q->region_bytes || q->jobs.size() >= 64) { int rcq = wgq_flush(); if (rcq != OARD_OK) return rcq; }
            float* part = (float*)(q->region + q->used);
            q->used += need;
            WgqJob J;
            memset(&J, 0, sizeof(J));
            J.P = Pm; J.Q = Qm; J.partial = part; J.bsum = db ? part + (size_t)p.n_chunks * p.PP * p.QP : nullptr; J.dW = dW; J.db = db;
            J.rows = rows; J.rpc = p.rpc; J.ldP = ldP; J.ncP = ncP; J.ldQ = ldQ; J.ncQ = ncQ; J.nPB = p.nPB; J.nQG = p.nQG;
            J.n_chunks = p.n_chunks; J.transposed = p.transposed; J.o_len = o_len; J.o_pad = o_pad; J.MO = MO; J.i_len = i_len;
            J.i_pad = i_pad; J.MI = MI; J.ldW = ldW; J.acc = acc; J.PP = p.PP; J.QP = p.QP;
            J.blk0 = q->blocks; J.rblk0 = q->rblocks;
            J.rblk_w = dW ? (unsigned)cdiv((long long)MO * MI, 32) : 0; J.rblk_b = db ? (unsigned)cdiv(MO, 4) : 0;
            q->blocks += (unsigned)(p.n_chunks * p.gy); q->rblocks += J.rblk_w + J.rblk_b;
            q->jobs.push_back(J);
            return OARD_OK;
        }
        ScopedLaunch sl_(F_WGRAD, st);
        const dim3 grid((unsigned)(p.n_chunks * p.gy)), block(256);
#define WG_LAUNCH(SILU_, NT_) hipLaunchKernelGGL((k_wgrad<SILU_, NT_>), grid, block, 0, st, Pm, ldP, ncP, Qm, ldQ, ncQ, 0LL, (long long)rows, \
                                                 p.rpc, p.nPB, p.nQG, partial, psum, qsum)
        if (p.NT == 7) { if (x_silu) WG_LAUNCH(true, 7); else WG_LAUNCH(false, 7); }
        else { if (x_silu) WG_LAUNCH(true, 8); else WG_LAUNCH(false, 8); }
#undef WG_LAUNCH
    }
    {
        ScopedLaunch sl_(F_WGRAD, st);
        if (dW)
            hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)cdiv((long long)MO * MI, 32)), dim3(256), 0, st, partial, p.n_chunks, p.PP,
                               p.QP, p.transposed, o_len, o_pad, MO, i_len, i_pad, MI, dW, ldW, acc);
        if (db)
            hipLaunchKernelGGL(k_bgrad_reduce, dim3((unsigned)cdiv(MO, 4)), dim3(256), 0, st, bpartial, p.n_chunks,
                               p.transposed ? p.QP : p.PP, o_len, o_pad, MO, db, acc);
    }
    HIP_TRY(hipGetLastError());
    return OARD_OK;
}

int oard_wgrad(const float* dY, int ldY, int ncY, int o_len, int o_pad, int MO, const float* X, int ldX, int ncX,
               int x_silu, int i_len, int i_pad, int MI, int64_t rows, float* dW, float* db, void* scratch,
               size_t scratch_bytes, oard_stream_t stream) {
    if (!dW) return OARD_EINVAL;
    return wgrad_impl(dY, ldY, ncY, o_len, o_pad, MO, X, ldX, ncX, x_silu, i_len, i_pad, MI, rows, dW, MI, db, 0, scratch, scratch_bytes,
                      (hipStream_t)stream);
}

}  // extern "C"
#include "oard_train_stages.h"
#include "oard_loss.h"
extern "C" {

// ---- the backward sweep through the C ABI (include/oard.h) ------------------------------------------------------------------------
size_t oard_train_scratch_bytes(const oard_config* c, const oard_topology* topo) {
    if (!config_ok(c) || !topo || topo->n_parts != 1) return 0;
    return make_train_tail(c, topo->parts[0].d, make_train_ws(c, topo->parts[0].d).total).total;
}

#define TRAIN_ENTER()                                                                                                          \
    if (!config_ok(c) || !topo || topo->n_parts != 1 || !packed || !packed_bwd || !tape || !scratch) return OARD_EINVAL;       \
    if (scratch_bytes < oard_train_scratch_bytes(c, topo)) return OARD_ENOMEM;                                                  \
    const TopoDev& tp = topo->parts[0].d;                                                                                      \
    const TrainCtx x(c, &tp, packed, packed_bwd, tape, scratch, params, grads, (hipStream_t)stream);                           \
    const TrainTail tw = make_train_tail(c, tp, x.w.total);                                                                    \
    (void)tw;                                                                                                                  \
    WgQueueScope wqs_((char*)scratch + x.w.wgq, x.w.wgq_bytes, x.stw);                                                          \
    int rc = OARD_EINVAL

int oard_train_scratch_poison(const oard_config* c, const oard_topology* topo, void* scratch, size_t scratch_bytes, oard_stream_t stream) {
    topo_touch(topo, (hipStream_t)stream);
    if (!scratch || scratch_bytes < oard_train_scratch_bytes(c, topo)) return OARD_EINVAL;
    HIP_TRY(hipMemsetAsync(scratch, 0xFF, oard_train_scratch_bytes(c, topo), (hipStream_t)stream));
    return OARD_OK;
}

int oard_train_layer_backward(const oard_config* c, const oard_topology* topo, const void* packed, const void* packed_bwd,
                              const void* tape, int layer, float* ds, float* dvec, float* dew, const float* const* params,
                              float* const* grads, void* scratch, size_t scratch_bytes, oard_stream_t stream) {
    topo_touch(topo, (hipStream_t)stream);
    TRAIN_ENTER();
    if (!ds || !dvec || !dew || layer < 0 || layer >= c->num_layers) return OARD_EINVAL;
    DISPATCH_DIMS(c, rc = tr_layer_bwd<D>(x, layer, ds, dvec, dew));
    if (rc == OARD_OK) rc = wgq_flush();
    return rc;
}

int oard_train_tail_backward(const oard_config* c, const oard_topology* topo, const void* packed, const void* packed_bwd,
                             const void* tape, const float* const* grad_out, float* ds, float* dvec, const float* const* params,
                             float* const* grads, void* scratch, size_t scratch_bytes, oard_stream_t stream) {
    topo_touch(topo, (hipStream_t)stream);
    TRAIN_ENTER();
    if (!ds || !dvec) return OARD_EINVAL;
    if (g_poison) HIP_TRY(hipMemsetAsync(scratch, 0xFF, oard_train_scratch_bytes(c, topo), (hipStream_t)stream));   // first call of a sweep
    DISPATCH_DIMS(c, rc = tr_tail_bwd<D>(x, tw, topo, grad_out, ds, dvec));
    if (rc == OARD_OK) rc = wgq_flush();
    x.join();                   // (a small stage; its operand buffers are reused by nothing until the next sweep, but tests read its gradients right away)
    return rc;
}

int oard_train_init_backward(const oard_config* c, const oard_topology* topo, const void* packed, const void* packed_bwd,
                             const void* tape, const float* const* xh, const float* ds0, const float* dew, const float* const* params,
                             float* const* grads, void* scratch, size_t scratch_bytes, oard_stream_t stream) {
    topo_touch(topo, (hipStream_t)stream);
    TRAIN_ENTER();
    if (!xh || !ds0 || !dew || !params) return OARD_EINVAL;
    DISPATCH_DIMS(c, rc = tr_init_bwd<D>(x, tw, topo, xh, ds0, dew));
    if (rc == OARD_OK) rc = wgq_flush();
    x.join();                   // end of the sweep: every gradient is complete in the caller's stream order
    return rc;
}

// one stage of a layer in isolation (teacher-forced tests; tests/test_grad_stages.py).  OARD_STAGE_RECOMPUTE must have run for the layer.
int oard_train_stage_backward(const oard_config* c, const oard_topology* topo, const void* packed, const void* packed_bwd,
                              const void* tape, int layer, int stage, const float* in0, const float* in1, const float* in2,
                              float* out0, float* out1, float* out2, const float* const* params, float* const* grads, void* scratch,
                              size_t scratch_bytes, oard_stream_t stream) {
    topo_touch(topo, (hipStream_t)stream);
    TRAIN_ENTER();
    if (layer < 0 || layer >= c->num_layers) return OARD_EINVAL;
    switch (stage) {
        case OARD_STAGE_RECOMPUTE: DISPATCH_DIMS(c, rc = tr_recompute<D>(x, layer)); break;
        case OARD_STAGE_UPDATE:    // in: ds, dvec  out: gs_a, gvec_a
            if (!in0 || !in1 || !out0 || !out1) return OARD_EINVAL;
            DISPATCH_DIMS(c, rc = tr_update_bwd<D>(x, layer, in0, in1, out0, out1)); break;
        case OARD_STAGE_MESSAGE:   // in: gs_a, gvec_a  out: gx, dxq, dvec_in   (d cd / d cr per inner edge stay in the scratch)
            if (!in0 || !in1 || !out0 || !out1 || !out2) return OARD_EINVAL;
            DISPATCH_DIMS(c, rc = tr_msg_bwd<D>(x, layer, in0, in1, out0, out1, out2)); break;
        case OARD_STAGE_GCL_NODE:  // in: gx, dxq  out: dxh, dagg
            if (!in0 || !in1 || !out0 || !out1) return OARD_EINVAL;
            DISPATCH_DIMS(c, rc = tr_gcl_node_bwd<D>(x, layer, in0, in1, out0, out1)); break;
        case OARD_STAGE_NODE_PRE:  // in: dxh, dP, dQ  out: ds_in
            if (!in0 || !in1 || !in2 || !out0) return OARD_EINVAL;
            DISPATCH_DIMS(c, rc = tr_pre_bwd<D>(x, layer, in0, in1, in2, out0)); break;
        case OARD_STAGE_GCL_EDGE: {  // in: dagg  in/out: out0 = dew  out: dP, dQ
            if (!in0 || !out0 || !out1 || !out2) return OARD_EINVAL;
            DISPATCH_DIMS(c, rc = tr_gcl_edge_bwd<D>(x, layer, in0, out0, out1, out2)); break; }
        case OARD_STAGE_EQUI_EDGE: { // in: dcd [A+1][3HP] (pads and spare row zero)  in/out: out0 = dew
            if (!in0 || !out0) return OARD_EINVAL;
            const RDims r(c->hidden, c->num_radial);
            HIP_TRY(hipMemcpyAsync(x.f(x.w.dcd), in0, (size_t)(tp.A + 1) * 3 * r.HP * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
            DISPATCH_DIMS(c, rc = tr_equi_edge_bwd<D>(x, layer, out0)); break; }
        default: return OARD_EINVAL;
    }
    if (rc == OARD_OK) rc = wgq_flush();
    x.join();                   // a test entry: its callers read the gradients (and poison / overwrite the scratch) in the caller's stream order
    return rc;
}

int oard_nan_replace(const oard_config* c, const oard_topology* topo, const int32_t* status, const float* const* noise,
                     float* const* out, oard_stream_t stream) {
    topo_touch(topo, (hipStream_t)stream);
    if (!config_ok(c) || !topo || !status || !noise || !out || c->n_obj != topo->n_obj) return OARD_EINVAL;
    NanPtrs np;
    memset(&np, 0, sizeof(np));
    for (int k = 0; k < c->n_obj; ++k) { np.noise[k] = noise[k]; np.out[k] = out[k]; np.node_nf[k] = c->node_nf[k]; }
    for (int p = 0; p < topo->n_parts; ++p)
        LAUNCH(F_OTHER, k_nan_replace, cdiv(topo->parts[p].d.N, 128), 128, (hipStream_t)stream, topo->parts[p].d, np, (const int*)status);
    HIP_TRY(hipGetLastError());
    return OARD_OK;
}

// ---- the training caller, fused (oard_loss.h) ------------------------------------------------------------------------------------
static LossCfg make_loss_cfg(const oard_config* c, const float* norm_values, const float* norm_biases, const float* scales, int pos_only,
                             int fixed_mask, int T) {
    LossCfg lc;
    memset(&lc, 0, sizeof(lc));
    for (int i = 0; i < 3; ++i) { lc.norm_value[i] = norm_values ? norm_values[i] : 1.0f; lc.norm_bias[i] = norm_biases ? norm_biases[i] : 0.0f; }
    for (int k = 0; k < c->n_obj; ++k) lc.scale[k] = scales ? scales[k] : 1.0f;
    lc.pos_only = pos_only; lc.fixed_mask = fixed_mask; lc.T = T;
    return lc;
}
int oard_loss_prepare(const oard_config* c, const oard_topology* topo, const float* const* pos, const int64_t* const* one_hot,
                      const int64_t* const* charge, const float* const* noise, const float* t_int, const float* gamma, int T,
                      const float* norm_values, const float* norm_biases, int pos_only, int fixed_mask, float* const* z,
                      float* const* eps, oard_stream_t stream) {
    topo_touch(topo, (hipStream_t)stream);
    if (!config_ok(c) || !topo || topo->n_parts != 1 || !pos || !one_hot || !charge || !noise || !t_int || !gamma || !z || !eps)
        return OARD_EINVAL;
    LossPtrs lp;
    memset(&lp, 0, sizeof(lp));
    for (int k = 0; k < c->n_obj; ++k) {
        if (c->node_nf[k] < 5 || c->node_nf[k] - 4 > 16) return OARD_EINVAL;      // [pos 3 | one_hot >= 1 | charge 1]
        lp.pos[k] = pos[k]; lp.one_hot[k] = (const long long*)one_hot[k]; lp.charge[k] = (const long long*)charge[k];
        lp.noise[k] = noise[k]; lp.z[k] = z[k]; lp.eps[k] = eps[k]; lp.node_nf[k] = c->node_nf[k];
    }
    const TopoDev& tp = topo->parts[0].d;
    LAUNCH(F_OTHER, k_loss_prep, cdiv(tp.N, 128), 128, (hipStream_t)stream, tp, lp,
           make_loss_cfg(c, norm_values, norm_biases, nullptr, pos_only, fixed_mask, T), t_int, gamma);
    HIP_TRY(hipGetLastError());
    return OARD_OK;
}
int oard_loss_terms(const oard_config* c, const oard_topology* topo, const float* const* eps, const float* const* net,
                    const float* const* z, const int64_t* const* one_hot, const int64_t* const* charge, const float* t_int,
                    const float* gamma, int T, const float* norm_values, const float* norm_biases, const float* scales, int pos_only,
                    int B, float* nll, float* terms, float* const* dnet, oard_stream_t stream) {
    topo_touch(topo, (hipStream_t)stream);
    if (!config_ok(c) || !topo || topo->n_parts != 1 || !eps || !net || !z || !one_hot || !charge || !t_int || !gamma || !nll || !terms ||
        !dnet || B < 1)
        return OARD_EINVAL;
    LossPtrs lp;
    memset(&lp, 0, sizeof(lp));
    for (int k = 0; k < c->n_obj; ++k) {
        if (c->node_nf[k] < 5 || c->node_nf[k] - 4 > 16) return OARD_EINVAL;
        lp.eps[k] = (float*)eps[k]; lp.net[k] = net[k]; lp.z[k] = (float*)z[k]; lp.one_hot[k] = (const long long*)one_hot[k];
        lp.charge[k] = (const long long*)charge[k]; lp.dnet[k] = dnet[k]; lp.node_nf[k] = c->node_nf[k];
    }
    const TopoDev& tp = topo->parts[0].d;
    LAUNCH(F_OTHER, k_loss_terms, tp.B, 64, (hipStream_t)stream, tp, lp, make_loss_cfg(c, norm_values, norm_biases, scales, pos_only, 0, T),
           t_int, gamma, B, nll, terms);
    HIP_TRY(hipGetLastError());
    return OARD_OK;
}
int oard_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* max_exp_avg_sq, int64_t n, double lr,
                    double beta1, double beta2, double eps, double weight_decay, int64_t step, int amsgrad, double grad_scale,
                    oard_stream_t stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || (amsgrad && !max_exp_avg_sq) || n < 0 || step < 1) return OARD_EINVAL;
    if (n == 0) return OARD_OK;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    LAUNCH(F_OTHER, k_adamw, cdiv(n, 256), 256, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, max_exp_avg_sq, (long long)n,
           (float)(1.0 - lr * weight_decay), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)(lr / bc1),
           (float)(1.0 / sqrt(bc2)), amsgrad, (float)grad_scale);
    HIP_TRY(hipGetLastError());
    return OARD_OK;
}

int oard_adamw_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* max_exp_avg_sq, int64_t n, double lr,
                        double beta1, double beta2, double eps, double weight_decay, int amsgrad, int clip, double* clip_state,
                        int capacity, const float* grad_norm, const float* flag, float* out4, oard_stream_t stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || (amsgrad && !max_exp_avg_sq) || n < 0 || !clip_state || capacity < 1 || capacity > 64 ||
        !grad_norm || !flag || !out4)
        return OARD_EINVAL;
    LAUNCH(F_OTHER, k_clip_decide, 1, 64, (hipStream_t)stream, clip_state, capacity, grad_norm, flag, clip, lr, beta1, beta2, eps, weight_decay, out4);
    if (n > 0)
        LAUNCH(F_OTHER, k_adamw_dev, cdiv(n, 256), 256, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, max_exp_avg_sq, (long long)n, amsgrad,
               reinterpret_cast<const AdamScal*>(clip_state + 4 + capacity));
    HIP_TRY(hipGetLastError());
    return OARD_OK;
}

// scratch buffers a test may want to look at (xq, cr, d cd, d cr of the last recompute / message stage)
int oard_train_scratch_entry(const oard_config* c, const oard_topology* topo, int which, size_t* offset_bytes, int64_t* rows,
                             int64_t* row_floats) {
    if (!config_ok(c) || !topo || topo->n_parts != 1 || !offset_bytes || !rows || !row_floats) return OARD_EINVAL;
    const TopoDev& d = topo->parts[0].d;
    const TrainWs w = make_train_ws(c, d);
    const RDims r(c->hidden, c->num_radial);
    switch (which) {
        case OARD_SCRATCH_XH: *offset_bytes = w.xh; *rows = d.N; *row_floats = r.HP; break;
        case OARD_SCRATCH_XQ: *offset_bytes = w.xq; *rows = d.N; *row_floats = 3 * r.HP; break;
        case OARD_SCRATCH_CR: *offset_bytes = w.cr; *rows = d.A + 1; *row_floats = 3 * r.HP; break;
        case OARD_SCRATCH_DCD: *offset_bytes = w.dcd; *rows = d.A + 1; *row_floats = 3 * r.HP; break;
        case OARD_SCRATCH_DCR: *offset_bytes = w.dcr; *rows = d.A + 1; *row_floats = 3 * r.HP; break;
        default: return OARD_EINVAL;
    }
    return OARD_OK;
}

int oard_tap(const oard_config* c, const oard_topology* topo, const void* ws_, int which, int layer, float* dst,
             oard_stream_t stream) {
    topo_touch(topo, (hipStream_t)stream);
    if (!config_ok(c) || !topo || !ws_ || !dst || layer != 0) return OARD_EINVAL;
    const RDims d(c->hidden, c->num_radial);
    ws_total(c, topo);
    hipStream_t st = (hipStream_t)stream;
    for (int p = 0; p < topo->n_parts; ++p) {
        const TopoDev& tp = topo->parts[p].d;
        const WsOff w = make_ws(c, tp);
        const char* ws = (const char*)ws_ + topo->parts[p].ws_off;
        auto nodes = [&](const float* src, int ld, int sections, int sect_pad, int sect_len) {
            const long long tot = (long long)tp.N * sections * sect_len;
            hipLaunchKernelGGL(k_tap_nodes, dim3((unsigned)cdiv(tot, 256)), dim3(256), 0, st, tp, src, ld, sections, sect_pad, sect_len, dst);
        };
        switch (which) {
            case OARD_TAP_S: nodes((const float*)(ws + w.s), d.HP, 1, d.HP, d.H); break;
            case OARD_TAP_VEC: nodes((const float*)(ws + topo->parts[p].vec_final), 3 * d.HP, 3, d.HP, d.H); break;
            case OARD_TAP_NE1: nodes((const float*)(ws + w.ne1), 3 * d.HP, 3, d.HP, d.H); break;
            case OARD_TAP_POS_FRAME: nodes((const float*)(ws + w.pf32), 3, 1, 3, 3); break;
            case OARD_TAP_DPOS: nodes((const float*)(ws + w.dpos), 3, 1, 3, 3); break;
            case OARD_TAP_HOUT: nodes((const float*)(ws + w.hout), 16, 1, 16, c->in_hidden); break;
            case OARD_TAP_LABELS:
                hipLaunchKernelGGL(k_tap_labels, dim3((unsigned)cdiv(tp.N, 256)), dim3(256), 0, st, tp, (const int*)(ws + w.labels), dst);
                break;
            case OARD_TAP_EDGE:
                if (tp.E > 0)
                    hipLaunchKernelGGL(k_tap_edges, dim3((unsigned)cdiv(tp.E * d.W, 256)), dim3(256), 0, st, tp,
                                       (const float*)(ws + w.ew), d.WP, d.W, dst);
                break;
            default: return OARD_EINVAL;
        }
    }
    HIP_TRY(hipGetLastError());
    return OARD_OK;
}

int oard_debug_stop_after(int code) { g_stop_after = code; return OARD_OK; }
int oard_debug_option(const char* name, int value) {
    if (!name) return OARD_EINVAL;
    if (!kV0 && value == 0 && (strcmp(name, "gcl_variant") == 0 || strcmp(name, "equi_variant") == 0 || strcmp(name, "node_variant") == 0))
        return OARD_EINVAL;                      // the first-generation kernels exist in experiment builds only
    if (strcmp(name, "gcl_variant") == 0) { g_gcl_variant = value; return OARD_OK; }
    if (strcmp(name, "equi_variant") == 0) { g_equi_variant = value; return OARD_OK; }
    if (strcmp(name, "node_variant") == 0) { g_node_variant = value; return OARD_OK; }
    if (strcmp(name, "experiments") == 0) return kV0 ? OARD_OK : OARD_EINVAL;      // query: is this an experiment build?
    if (strcmp(name, "gcl_skip") == 0) { g_gcl_skip = value; return OARD_OK; }
    if (strcmp(name, "equi_skip") == 0) { g_equi_skip = value; return OARD_OK; }
    if (strcmp(name, "gcl_persist") == 0) { g_gcl_persist = value; return OARD_OK; }
    if (strcmp(name, "gcl_grid") == 0) { g_gcl_grid = value; return OARD_OK; }
    if (strcmp(name, "skip_families") == 0) { g_skip_families = value; return OARD_OK; }
    if (strcmp(name, "parts") == 0) { g_parts = value; return OARD_OK; }
    if (strcmp(name, "sequential") == 0) { g_sequential = value; return OARD_OK; }
    if (strcmp(name, "poison") == 0) { g_poison = value; return OARD_OK; }
    if (strcmp(name, "auto_small") == 0) { g_auto_small = value; return OARD_OK; }
    if (strcmp(name, "auto_tiny") == 0) { g_auto_tiny = value; return OARD_OK; }
    if (strcmp(name, "npb") == 0) { g_npb = value; return OARD_OK; }
    if (strcmp(name, "wgrad_wgs") == 0) { g_wgrad_wgs = value; return OARD_OK; }
    if (strcmp(name, "wgrad_t16") == 0) { g_wgrad_t16 = value; return OARD_OK; }
    if (strcmp(name, "wgrad_queue") == 0) { g_wgrad_queue = value; return OARD_OK; }
    if (strcmp(name, "gate_fold") == 0) { g_gate_fold = value; return OARD_OK; }
    if (strcmp(name, "wgrad_lds") == 0) { g_wgrad_lds = value; return OARD_OK; }
    if (strcmp(name, "wgrad_shapes") == 0) { g_wgrad_shapes = value & 3; return OARD_OK; }
    if (strcmp(name, "train_dual") == 0) { g_train_dual = value; return OARD_OK; }
    if (strcmp(name, "small_split") == 0) { g_small_split = value; return OARD_OK; }
    return OARD_EINVAL;
}
#ifdef OARD_PHASE_PROBE
// experiment build: read (and clear) the cycle sums of the GCL phase probe (oard_edge_v1.h); not declared in include/oard.h
int oard_debug_probe_read(unsigned long long* out8) {
    if (!out8) return OARD_EINVAL;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_phase_probe), 8 * sizeof(unsigned long long)));
    unsigned long long z[8] = {};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_probe), z, sizeof(z)));
    return OARD_OK;
}
#endif
#ifdef OARD_TIMELINE
int oard_debug_timeline_read(long long* out, int waves) {          // experiment build: [waves][TL_MAX] (oard_edge_v1.h)
    if (!out || waves < 1 || waves > 16) return OARD_EINVAL;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_timeline), (size_t)waves * TL_MAX * sizeof(long long)));
    return OARD_OK;
}
#endif
#ifdef OARD_EXPERIMENTS
// number of bounded waits of the flag-pipelined kernels that ran out (must be 0; not declared in include/oard.h: debugging aid)
int oard_debug_fp_timeouts(unsigned int* out) {
    if (!out) return OARD_EINVAL;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fp_timeouts), sizeof(unsigned int)));
    return OARD_OK;
}
#endif
int oard_timing_enable(int on) { g_timing.on = on != 0; return OARD_OK; }
int oard_timing_reset(void) {
    g_timing.flush();
    for (int f = 0; f < F_COUNT; ++f) { g_timing.total_ms[f] = 0; g_timing.launches[f] = 0; }
    return OARD_OK;
}
int oard_timing_get(const char* family, double* total_ms, int64_t* launches) {
    if (!family || !total_ms || !launches) return OARD_EINVAL;
    g_timing.flush();
    for (int f = 0; f < F_COUNT; ++f)
        if (strcmp(family, kFamilyNames[f]) == 0) { *total_ms = g_timing.total_ms[f]; *launches = g_timing.launches[f]; return OARD_OK; }
    return OARD_EINVAL;
}

}  // extern "C"
