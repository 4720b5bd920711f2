// Host executor for the general-edge-list path (oareactdiff_amd/csrc/oard_general.h): the SAME stage functors and the same
// orchestration run in plain loops on host pointers.  TEST INFRASTRUCTURE (tests/test_general_host.py compiles it with g++ and checks the
// formulas against the oracle / the reference's goldens where no GPU exists); the product entry points live in the HIP library only.
#include <algorithm>
#include <cstdlib>

#include "../../oareactdiff_amd/csrc/oard_general.h"

namespace og = oard_general;

struct HostExec {
    template <class F> int run(long long n, const F& f) { for (long long i = 0; i < n; ++i) f(i); return OARD_OK; }
    int zero(void* p, size_t bytes) { memset(p, 0, bytes); return OARD_OK; }
    int gemm(const og::Gemm& k) { return run(k.threads(), k); }
};

extern "C" int oard_general_forward_host(const oard_config* c, const int64_t* cm, const int64_t* nfs, int64_t N, const int64_t* ei, int64_t E,
                                         const float* const* params, size_t n_params, const float* const* xh, const float* t, int t_scalar,
                                         const float* cond, float* const* out, int32_t* status) {
    og::GraphHost gh;
    int rc = og::build_graph(c, cm, nfs, N, ei, E, gh);
    if (rc != OARD_OK) return rc;
    if (n_params != (size_t)og::Params(c).count) return OARD_EINVAL;
    og::Graph g{gh.N, gh.E, gh.G, gh.ei0.data(), gh.ei1.data(), gh.in_ptr.data(), gh.in_list.data(), gh.out_ptr.data(), gh.out_list.data(),
                gh.sub.data(), gh.node_obj.data(), gh.node_row.data(), gh.node_tidx.data(), gh.node_grp.data(), gh.grp_ptr.data(),
                gh.grp_list.data()};
    const size_t bytes = og::carve(c, N, E, gh.G, nullptr).bytes;
    char* base = (char*)malloc(bytes ? bytes : 1);
    if (!base) return OARD_ENOMEM;
    memset(base, 0xff, bytes);                                 // NaN patterns: nothing may depend on the workspace's contents
    const og::Workspace w = og::carve(c, N, E, gh.G, base);
    HostExec ex;
    rc = og::forward(ex, c, g, params, params, xh, t, t_scalar, cond, out, w, status);
    free(base);
    return rc;
}
