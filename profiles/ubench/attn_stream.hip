// attn_stream.hip -- the causal attention kernels at SASRec's shape (4096 sequences, L = 200, d = 64) timed alone and checked
// against a float64 loop on sampled sequences; RBX_ATTN_STREAM=0 / 1 selects the resident / the streamed form
// (rbx_attn_mfma.hip / rbx_attn_stream.h).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include profiles/ubench/attn_stream.hip -o profiles/ubench/attn_stream
//   ./attn_stream [L] [BH] [p_drop_is_ignored]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../recbox_amd/csrc/rbx_common.hip"
#include "../../recbox_amd/csrc/rbx_attn_mfma.hip"

int main(int argc, char** argv) {
  const int L = argc > 1 ? atoi(argv[1]) : 200, BH = argc > 2 ? atoi(argv[2]) : 4096, HD = 64;
  const bool bwd = argc > 3 && atoi(argv[3]) != 0;
  const size_t n = static_cast<size_t>(BH) * L * HD;
  std::vector<float> hq(n), hk(n), hv(n), hg(n);
  unsigned x = 12345u;
  auto rnd = [&]() { x = x * 1664525u + 1013904223u; return static_cast<float>(x >> 8) / 8388608.f - 1.f; };
  for (size_t i = 0; i < n; ++i) { hq[i] = rnd(); hk[i] = rnd(); hv[i] = rnd(); hg[i] = rnd(); }
  float *q, *k, *v, *o, *lse, *g, *dq, *dk, *dv, *scr;
  (void)hipMalloc(&q, n * 4); (void)hipMalloc(&k, n * 4); (void)hipMalloc(&v, n * 4); (void)hipMalloc(&o, n * 4);
  (void)hipMalloc(&g, n * 4); (void)hipMalloc(&dq, n * 4); (void)hipMalloc(&dk, n * 4); (void)hipMalloc(&dv, n * 4);
  (void)hipMalloc(&lse, static_cast<size_t>(BH) * L * 4);
  (void)hipMalloc(&scr, static_cast<size_t>(BH) * L * 4);
  (void)hipMemcpy(q, hq.data(), n * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(k, hk.data(), n * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(v, hv.data(), n * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(g, hg.data(), n * 4, hipMemcpyHostToDevice);
  (void)hipMemset(o, 0xff, n * 4);
  rbx::DropArgs drop{};
  const float scale = 0.125f;
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  auto fwd = [&]() { return rbx::attn_mfma_fwd(q, k, v, BH, L, HD, scale, 1, o, lse, drop, nullptr); };
  auto back = [&]() { return rbx::attn_mfma_bwd(q, k, v, o, g, lse, BH, L, HD, scale, 1, dq, dk, dv, scr, drop, nullptr); };
  for (int i = 0; i < 3; ++i) { fwd(); if (bwd) back(); }
  (void)hipDeviceSynchronize();
  printf("launch: %s\n", hipGetErrorString(hipGetLastError()));
  const int reps = 20;
  float ms = 0.f;
  (void)hipEventRecord(a);
  for (int i = 0; i < reps; ++i) fwd();
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  (void)hipEventElapsedTime(&ms, a, b);
  const char* form = getenv("RBX_ATTN_STREAM");
  printf("STREAM=%s L=%d BH=%d forward %.1f us\n", form ? form : "(default)", L, BH, ms * 1000.f / reps);
  if (bwd) {
    (void)hipEventRecord(a);
    for (int i = 0; i < reps; ++i) back();
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    (void)hipEventElapsedTime(&ms, a, b);
    printf("STREAM=%s L=%d BH=%d backward %.1f us\n", form ? form : "(default)", L, BH, ms * 1000.f / reps);
  }
  // float64 check on sampled sequences
  std::vector<float> ho(n), hl(static_cast<size_t>(BH) * L), hdq, hdk, hdv;
  (void)hipMemcpy(ho.data(), o, n * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(hl.data(), lse, hl.size() * 4, hipMemcpyDeviceToHost);
  if (bwd) {
    hdq.resize(n); hdk.resize(n); hdv.resize(n);
    (void)hipMemcpy(hdq.data(), dq, n * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hdk.data(), dk, n * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hdv.data(), dv, n * 4, hipMemcpyDeviceToHost);
  }
  double eo = 0, el = 0, eq = 0, ek = 0, ev = 0;
  const int picks[6] = {0, 1, 2, BH / 2 + 1, BH - 2, BH - 1};
  for (int pi = 0; pi < 6; ++pi) {
    const size_t bh = picks[pi], base = bh * L * HD;
    std::vector<double> P(static_cast<size_t>(L) * L, 0.0), O(static_cast<size_t>(L) * HD, 0.0);
    for (int i = 0; i < L; ++i) {
      std::vector<double> s(i + 1);
      double mx = -1e300;
      for (int j = 0; j <= i; ++j) {
        double d = 0;
        for (int c = 0; c < HD; ++c) d += static_cast<double>(hq[base + i * HD + c]) * hk[base + j * HD + c];
        s[j] = d * scale;
        mx = std::max(mx, s[j]);
      }
      double sum = 0;
      for (int j = 0; j <= i; ++j) sum += std::exp(s[j] - mx);
      for (int j = 0; j <= i; ++j) P[static_cast<size_t>(i) * L + j] = std::exp(s[j] - mx) / sum;
      for (int c = 0; c < HD; ++c) {
        double acc = 0;
        for (int j = 0; j <= i; ++j) acc += P[static_cast<size_t>(i) * L + j] * hv[base + j * HD + c];
        O[i * HD + c] = acc;
        eo = std::max(eo, std::fabs(acc - ho[base + i * HD + c]));
      }
      el = std::max(el, std::fabs(mx + std::log(sum) - hl[bh * L + i]));
    }
    if (bwd) {
      std::vector<double> dS(static_cast<size_t>(L) * L, 0.0);
      for (int i = 0; i < L; ++i) {
        double D = 0;
        for (int c = 0; c < HD; ++c) D += static_cast<double>(hg[base + i * HD + c]) * O[i * HD + c];
        for (int j = 0; j <= i; ++j) {
          double dp = 0;
          for (int c = 0; c < HD; ++c) dp += static_cast<double>(hg[base + i * HD + c]) * hv[base + j * HD + c];
          dS[static_cast<size_t>(i) * L + j] = P[static_cast<size_t>(i) * L + j] * (dp - D);
        }
      }
      for (int i = 0; i < L; ++i)
        for (int c = 0; c < HD; ++c) {
          double aq = 0, ak = 0, av = 0;
          for (int j = 0; j <= i; ++j) aq += dS[static_cast<size_t>(i) * L + j] * hk[base + j * HD + c];
          for (int j = i; j < L; ++j) {
            ak += dS[static_cast<size_t>(j) * L + i] * hq[base + j * HD + c];
            av += P[static_cast<size_t>(j) * L + i] * hg[base + j * HD + c];
          }
          eq = std::max(eq, std::fabs(aq * scale - hdq[base + i * HD + c]));
          ek = std::max(ek, std::fabs(ak * scale - hdk[base + i * HD + c]));
          ev = std::max(ev, std::fabs(av - hdv[base + i * HD + c]));
        }
    }
  }
  printf("max |error| vs float64 on 6 sequences: O %.3g  LSE %.3g", eo, el);
  if (bwd) printf("  dQ %.3g  dK %.3g  dV %.3g", eq, ek, ev);
  printf("\n");
  return 0;
}
