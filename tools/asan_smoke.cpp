// Host-side AddressSanitizer smoke of the C ABI (make -C dynesty_amd/csrc asan, then:
//   hipcc -fsanitize=address -shared-libsan -O1 -g -I include tools/asan_smoke.cpp -L dynesty_amd -ldynhip_asan \
//         -Wl,-rpath,$PWD/dynesty_amd -o tools/asan_smoke && HSA_XNACK=1 ASAN_OPTIONS=detect_leaks=0 tools/asan_smoke
// exercises create / rebuild (single, multi) / contains / bootstrap expansion / destroy on host buffers sized exactly
// as include/dynhip.h documents, so an out-of-bounds copy-out of the library trips the sanitizer.
#include <cstdio>
#include <cstdint>
#include <vector>
#include "dynhip.h"

int main() {
  dh_ctx* ctx = dh_create(0);
  if (!ctx) { std::printf("no device: %s\n", dh_last_error(nullptr)); return 2; }
  const int n = 600, d = 3;
  std::vector<double> pts((size_t)n * d);
  unsigned long long x = 88172645463325252ull;
  for (size_t i = 0; i < pts.size(); ++i) {
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    pts[i] = 0.3 + 0.4 * (double)(x >> 11) / 9007199254740992.0 + ((i / d) % 2 ? 0.2 : 0.0);
  }
  int rc = 0;
  for (int mode = 0; mode < 2; ++mode) {
    const int me = mode == 0 ? n / (2 * d) : 1;
    int32_t nells = 0, nnodes = 0;
    std::vector<double> ctrs((size_t)me * d), covs((size_t)me * d * d), ams(covs.size()), axes(covs.size()),
        axl((size_t)me * d), lv(me);
    std::vector<int32_t> leaf(n);
    rc = dh_rebuild(ctx, pts.data(), n, d, mode, me, &nells, ctrs.data(), covs.data(), ams.data(), axes.data(), axl.data(),
                    lv.data(), leaf.data(), &nnodes);
    std::printf("rebuild mode %d: rc %d, %d ellipsoids\n", mode, rc, (int)nells);
    if (rc) return 1;
    std::vector<int32_t> count(n);
    std::vector<uint64_t> mask((size_t)n * ((nells + 63) / 64));
    std::vector<double> quad((size_t)n * nells);
    rc = dh_contains(ctx, pts.data(), n, d, ctrs.data(), ams.data(), nells, 0, count.data(), mask.data(), quad.data());
    std::printf("contains: rc %d\n", rc);
    if (rc) return 1;
    const uint64_t ent[4] = {1, 2, 3, 4};
    double expand = 0.0;
    std::vector<int32_t> nin(5);
    rc = dh_bootstrap_expand(ctx, 1, pts.data(), n, d, mode == 0, 5, ent, &expand, nin.data());
    std::printf("bootstrap_expand: rc %d, expand %.6f\n", rc, expand);
    if (rc) return 1;
  }
  // the device-resident loop on the 3-D unit Normal (prior +-10): unif + bootstrap 5, and rwalk with per-point output
  {
    const double like_par[1] = {-2.756815599614018};  // -1.5 ln(2 pi)
    const double prior_par[2] = {10.0, 0.0};
    const int prob = dh_problem_create(ctx, 3, 0 /* iid Normal */, like_par, 1, 1 /* affine */, prior_par, 2);
    if (prob < 0) return 1;
    const int runs = 3, nlive = 200, K = 48;
    const int64_t max_iter = 20000;
    const uint32_t words[2] = {7u, 11u};
    for (int sampler : {6, 0}) {
      std::vector<double> rec((size_t)runs * 8), dead((size_t)runs * max_iter), live((size_t)runs * nlive),
          dead_u((size_t)runs * max_iter * 3), live_u((size_t)runs * nlive * 3);
      std::vector<int32_t> did((size_t)runs * max_iter), dit(did.size()), dnc(did.size()), lit((size_t)runs * nlive);
      int64_t nf = 0;
      rc = dh_ns_ensemble(ctx, prob, runs, nlive, 3, K, sampler, 23, sampler == 0, 0, 0.1, sampler == 6 ? 1.0 : 1.25, 0,
                          max_iter, words, 2, 0, rec.data(), dead.data(), live.data(), dead_u.data(), live_u.data(), &nf,
                          did.data(), dit.data(), dnc.data(), lit.data(), sampler == 6 ? 5 : 0, 0);
      std::printf("ns_ensemble sampler %d: rc %d, %lld fills, ln Z %.3f %.3f %.3f (truth -8.987)\n", sampler, rc,
                  (long long)nf, rec[0], rec[8], rec[16]);
      if (rc) return 1;
    }
    dh_problem_destroy(ctx, prob);
  }
  dh_destroy(ctx);
  std::printf("asan smoke done\n");
  return 0;
}
