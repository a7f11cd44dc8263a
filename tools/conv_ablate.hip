// Timing-only ablation of the fp32 MFMA conv kernel (not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I image_restoration_amd/csrc tools/conv_ablate.hip \
//         image_restoration_amd/csrc/capi.hip -o gpurun_out/conv_ablate && gpurun_out/conv_ablate
#include "../image_restoration_amd/csrc/conv_f32.hip"

#include <vector>

template <int COT, int PT, int ABL, int NW = 4>
float run(const ConvParams& p, int n, int iters, hipStream_t st) {
  constexpr int lds = conv_lds_bytes<COT, PT, 3, NW>();
  auto kern = conv_f32_kernel<COT, PT, 3, false, ABL, NW>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  dim3 grid(p.tiles_x * p.tiles_y * n, 1);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, st, p);
  hipEventRecord(a, st);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, st, p);
  hipEventRecord(b, st);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms / iters;
}

int main() {
  const int N = 16, H = 128, W = 128;
  hipStream_t st;
  hipStreamCreate(&st);
  for (int cin : {64, 128, 192}) {
    for (int cot : {1, 2}) {
      const int cout = 32 * cot;
      float *in, *w, *out;
      hipMalloc(&in, (size_t)N * cin * H * W * 4);
      hipMalloc(&out, (size_t)N * cout * H * W * 4);
      hipMalloc(&w, (size_t)cout * cin * 9 * 4);
      hipMemset(in, 0x3c, (size_t)N * cin * H * W * 4);  // ~0.011 floats
      hipMemset(w, 0x3c, (size_t)cout * cin * 9 * 4);
      ConvParams p = {};
      p.zero = sr::zero_line();
      p.in = in; p.w = w; p.out = out;
      p.in_ns = (long long)cin * H * W; p.out_ns = (long long)cout * H * W;
      p.cin_blocks = cin / 8; p.cout_blocks = cout / 8; p.cout = cout;
      p.in_h = H; p.in_w = W; p.H = p.vH = p.oH = H; p.W = p.vW = p.oW = W;
      p.src_mul = 1; p.dst_mul = 1; p.tap_oy = p.tap_ox = -1;
      p.tiles_x = W / 32; p.res_cb1 = 1 << 30; p.slope = 0.2f; p.alpha = 1.f;
      const double flops = 2.0 * 9 * cin * cout * (double)N * H * W;
      auto report = [&](const char* name, float ms) { printf("cin %3d cout %2d %-28s %8.1f us  %6.1f TF/s\n", cin, cout, name, ms * 1e3, flops / ms / 1e9); };
      p.tiles_y = H / 8;
      if (cot == 1) {
        report("PT2 full", run<1, 2, 0>(p, N, 20, st));
        report("PT2 no-refill", run<1, 2, 1>(p, N, 20, st));
        report("PT2 no-barrier", run<1, 2, 2>(p, N, 20, st));
        report("PT2 no-refill no-barrier", run<1, 2, 3>(p, N, 20, st));
        report("PT2 no-store", run<1, 2, 8>(p, N, 20, st));
        report("PT2 no-refill no-store", run<1, 2, 9>(p, N, 20, st));
        p.tiles_y = H / 16;
        report("PT4 full", run<1, 4, 0>(p, N, 20, st));
        report("PT4 no-refill no-barrier", run<1, 4, 3>(p, N, 20, st));
        report("8 waves PT2 full", run<1, 2, 0, 8>(p, N, 20, st));
        p.tiles_y = H / 32;
        report("8 waves PT4 full", run<1, 4, 0, 8>(p, N, 20, st));
      } else {
        report("PT2 full", run<2, 2, 0>(p, N, 20, st));
        report("PT2 no-refill", run<2, 2, 1>(p, N, 20, st));
        report("PT2 no-barrier", run<2, 2, 2>(p, N, 20, st));
        report("PT2 no-refill no-barrier", run<2, 2, 3>(p, N, 20, st));
        report("PT2 no-store", run<2, 2, 8>(p, N, 20, st));
        report("PT2 no-refill no-store", run<2, 2, 9>(p, N, 20, st));
        p.tiles_y = H / 16;
        report("PT4 full", run<2, 4, 0>(p, N, 20, st));
        report("PT4 no-refill no-barrier", run<2, 4, 3>(p, N, 20, st));
        report("8 waves PT2 full", run<2, 2, 0, 8>(p, N, 20, st));
        report("8 waves PT2 no-refill", run<2, 2, 1, 8>(p, N, 20, st));
      }
      hipFree(in); hipFree(out); hipFree(w);
    }
  }
  return 0;
}
