// wgbench.cpp -- timing of the grouped LoKr weight-gradient launches (lyc_lokr_wgrad_group) over the SDXL layer mix, with the
// library source compiled IN (so that the LYC_WG_* plan constants of capi.hip can be varied per binary):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ilycoris_amd/csrc -DLYC_WG_TARGET=64 benchmarks/wgbench.cpp -o wgbench_t64 -lrocblas
// Prints ms per call and algorithmic GB/s (g + x read once) per shape class and for the whole SDXL Linear mix.
#include "../lycoris_amd/csrc/capi.hip"

#include <cstdio>
#include <random>
#include <vector>

struct Shape {
  long M;
  int a, c, d, count;
  const char* tag;
};

#define CK(x)                                                              \
  do {                                                                     \
    hipError_t e_ = (x);                                                   \
    if (e_ != hipSuccess) {                                                \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));              \
      return 1;                                                            \
    }                                                                      \
  } while (0)

int main() {
  const std::vector<Shape> sdxl = {{1024, 8, 160, 160, 372, "attn/proj @1280"}, {1024, 8, 1280, 160, 60, "ff.net.0 @1280"},
                                   {1024, 8, 160, 640, 60, "ff.net.2 @1280"},  {4096, 8, 80, 80, 70, "attn/proj @640"},
                                   {4096, 8, 640, 80, 10, "ff.net.0 @640"},    {4096, 8, 80, 320, 10, "ff.net.2 @640"},
                                   {77, 8, 160, 256, 120, "attn2 k/v @1280"},  {77, 8, 80, 256, 20, "attn2 k/v @640"}};
  std::vector<LycLokrWgradItem> all;
  std::vector<std::pair<size_t, size_t>> range;  // items of each shape class
  std::mt19937 rng(7);
  std::vector<unsigned short> host;
  double total_bytes = 0;
  std::vector<double> class_bytes;
  for (const Shape& s : sdxl) {
    const size_t lo = all.size();
    const size_t ng = (size_t)s.M * s.a * s.c, nx = (size_t)s.M * s.a * s.d;
    for (int k = 0; k < s.count; ++k) {
      void *g, *x;
      float *w1, *dw2;
      CK(hipMalloc(&g, ng * 2));
      CK(hipMalloc(&x, nx * 2));
      CK(hipMalloc((void**)&w1, 64 * 4));
      CK(hipMalloc((void**)&dw2, (size_t)s.c * s.d * 4));
      host.resize(ng > nx ? ng : nx);
      for (auto& v : host) v = (unsigned short)(0x3c00 + (rng() & 0x1ff));  // bf16 values around 0.01
      CK(hipMemcpy(g, host.data(), ng * 2, hipMemcpyHostToDevice));
      CK(hipMemcpy(x, host.data(), nx * 2, hipMemcpyHostToDevice));
      CK(hipMemset(w1, 0, 64 * 4));
      CK(hipMemset(dw2, 0, (size_t)s.c * s.d * 4));
      all.push_back(LycLokrWgradItem{g, x, w1, nullptr, dw2, nullptr, s.M, s.a, s.a, s.c, s.d, 1.0f});
    }
    range.push_back({lo, all.size()});
    class_bytes.push_back((double)s.count * 2.0 * (ng + nx));
    total_bytes += class_bytes.back();
  }
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto time_call = [&](const LycLokrWgradItem* it, int n, float& ms) -> int {
    const int reps = 5;
    if (lyc_lokr_wgrad_group(it, n, LYC_BF16, st)) { fprintf(stderr, "%s\n", lyc_last_error()); return 1; }
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) lyc_lokr_wgrad_group(it, n, LYC_BF16, st);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    return 0;
  };
  printf("plan: target %d  maxrows %d  big_rows %d  U %d  wide %d\n", LYC_WG_TARGET, LYC_WG_MAXROWS, LYC_WG_BIG_ROWS, LYC_WG_U,
         LYC_WG_WIDE);
  for (size_t c = 0; c < sdxl.size(); ++c) {
    float ms = 0;
    if (time_call(all.data() + range[c].first, (int)(range[c].second - range[c].first), ms)) return 1;
    KronDw2sArgs da{};
    da.M = sdxl[c].M; da.G = sdxl[c].a; da.I = sdxl[c].c; da.J = sdxl[c].d;
    const int cfg = plan_dw2s(da, true);
    printf("%-18s M %5ld c %4d d %4d x%3d : %8.3f ms  %6.2f us/problem  %7.1f GB/s   tile %s  slabs %d  rows/slab %ld\n", sdxl[c].tag,
           sdxl[c].M, sdxl[c].c, sdxl[c].d, sdxl[c].count, ms, ms * 1e3 / sdxl[c].count, class_bytes[c] / (ms * 1e-3) / 1e9,
           cfg == DW2_T44 ? "64x64" : cfg == DW2_T52 ? "80x32" : "32x32", da.nsplit, da.rows_per_block);
  }
  float ms = 0;
  if (time_call(all.data(), (int)all.size(), ms)) return 1;
  printf("SDXL Linear mix, %zu problems in one call: %8.3f ms  %7.1f GB/s\n", all.size(), ms, total_bytes / (ms * 1e-3) / 1e9);

  // ---- LoCon (rank 16): the same layers as [M, I] x / [M, O] g with I = 8 d, O = 8 c ------------------------------------------
  printf("locon plan: rows/slab %d  cvmax %d\n", LYC_TNG_ROWS, LYC_TNG_CVMAX);
  std::vector<LycLoconWgradItem> lall;
  std::vector<std::pair<size_t, size_t>> lrange;
  for (size_t c = 0; c < sdxl.size(); ++c) {
    const Shape& s = sdxl[c];
    const size_t lo = lall.size();
    const int I = s.a * s.d, O = s.a * s.c, r = 16;
    for (size_t k = range[c].first; k < range[c].second; ++k) {
      float *t, *dt, *dd, *du;
      CK(hipMalloc((void**)&t, (size_t)s.M * r * 4));
      CK(hipMalloc((void**)&dt, (size_t)s.M * r * 4));
      CK(hipMalloc((void**)&dd, (size_t)r * I * 4));
      CK(hipMalloc((void**)&du, (size_t)r * O * 4));
      CK(hipMemset(t, 0, (size_t)s.M * r * 4));
      CK(hipMemset(dt, 0, (size_t)s.M * r * 4));
      CK(hipMemset(dd, 0, (size_t)r * I * 4));
      CK(hipMemset(du, 0, (size_t)r * O * 4));
      lall.push_back(LycLoconWgradItem{all[k].g, all[k].x, t, dt, dd, du, s.M, I, O, r, 1.0f});
    }
    lrange.push_back({lo, lall.size()});
  }
  auto time_locon = [&](const LycLoconWgradItem* it, int n, float& ms_) -> int {
    const int reps = 5;
    if (lyc_locon_wgrad_group(it, n, LYC_BF16, st)) { fprintf(stderr, "%s\n", lyc_last_error()); return 1; }
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) lyc_locon_wgrad_group(it, n, LYC_BF16, st);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms_, e0, e1));
    ms_ /= reps;
    return 0;
  };
  for (size_t c = 0; c < sdxl.size(); ++c) {
    if (time_locon(lall.data() + lrange[c].first, (int)(lrange[c].second - lrange[c].first), ms)) return 1;
    printf("locon %-18s x%3d : %8.3f ms  %6.2f us/problem  %7.1f GB/s\n", sdxl[c].tag, sdxl[c].count, ms, ms * 1e3 / sdxl[c].count,
           class_bytes[c] / (ms * 1e-3) / 1e9);
  }
  if (time_locon(lall.data(), (int)lall.size(), ms)) return 1;
  printf("locon SDXL Linear mix, %zu problems in one call: %8.3f ms  %7.1f GB/s\n", lall.size(), ms, total_bytes / (ms * 1e-3) / 1e9);

  // ---- LoHa (rank 32): G = g^T x per layer (library GEMM) + grouped HadaWeight.backward ----------------------------------------
  printf("loha plan: target %d  ntmax %d\n", LYC_LHG_TARGET, LYC_LHG_NTMAX);
  std::vector<LycLohaWgradItem> hall;
  std::vector<std::pair<size_t, size_t>> hrange;
  for (size_t c = 0; c < sdxl.size(); ++c) {
    const Shape& s = sdxl[c];
    const size_t lo = hall.size();
    const int I = s.a * s.d, O = s.a * s.c, r = 32;
    for (size_t k = range[c].first; k < range[c].second; ++k) {
      float *f[4], *d[4], *gw;
      const size_t na = (size_t)O * r, nb = (size_t)r * I;
      for (int j = 0; j < 4; ++j) {
        const size_t nf = (j & 1) ? nb : na;
        CK(hipMalloc((void**)&f[j], nf * 4));
        CK(hipMalloc((void**)&d[j], nf * 4));
        CK(hipMemset(f[j], 0, nf * 4));
        CK(hipMemset(d[j], 0, nf * 4));
      }
      CK(hipMalloc((void**)&gw, (size_t)O * I * 4));
      hall.push_back(LycLohaWgradItem{all[k].g, all[k].x, f[0], f[1], f[2], f[3], d[0], d[1], d[2], d[3], gw, s.M, I, O, r, 1.0f});
    }
    hrange.push_back({lo, hall.size()});
  }
  auto time_loha = [&](const LycLohaWgradItem* it, int n, float& ms_) -> int {
    const int reps = 3;
    if (lyc_loha_wgrad_group(it, n, LYC_BF16, st)) { fprintf(stderr, "%s\n", lyc_last_error()); return 1; }
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) lyc_loha_wgrad_group(it, n, LYC_BF16, st);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms_, e0, e1));
    ms_ /= reps;
    return 0;
  };
  for (size_t c = 0; c < sdxl.size(); ++c) {
    if (time_loha(hall.data() + hrange[c].first, (int)(hrange[c].second - hrange[c].first), ms)) return 1;
    int no = 1, nt = 1;
    plan_loha_grad(sdxl[c].a * sdxl[c].c, sdxl[c].a * sdxl[c].d, 32, true, no, nt);
    printf("loha %-18s x%3d : %8.3f ms  %6.2f us/problem   block %d x %d\n", sdxl[c].tag, sdxl[c].count, ms, ms * 1e3 / sdxl[c].count, no, nt);
  }
  if (time_loha(hall.data(), (int)hall.size(), ms)) return 1;
  printf("loha SDXL Linear mix, %zu problems in one call (G GEMMs + factor gradients): %8.3f ms\n", hall.size(), ms);
  return 0;
}
