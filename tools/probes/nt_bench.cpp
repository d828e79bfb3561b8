// Stand-alone A/B bench of the implicit-GEMM kernels through the C ABI of libvc_hip.so (no Python, no torch):
// every layer is run with the 8-wave 256-wide kernels (VC_NT8=1 / VC_WGRAD8=1) and the 4-wave 128-wide ones (=0), outputs
// are compared with each other and (NTB_NAIVE=1) with the plain-FMA kernels, and each launch is timed with HIP events on
// cold operands (a 1 GiB fill runs before every timed launch).  NTB_PASS selects 0 forward, 1 data gradient, 2 weight
// gradient (default: all three).
//   hipcc -O2 --offload-arch=gfx950 tools/probes/nt_bench.cpp -o tools/probes/nt_bench.bin -Lvideo-classification_amd -lvc_hip
//   LD_LIBRARY_PATH=video-classification_amd tools/probes/nt_bench.bin [layer-name-substring ...]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <string>

#include "../../include/vc_hip.h"

#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

__global__ void fill_bf16(uint16_t* p, size_t n, uint32_t seed, float scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    const float v = ((float)(h & 0xFFFFFF) / 8388608.0f - 1.0f) * scale;  // uniform [-scale, scale)
    __bf16 b = (__bf16)v;
    p[i] = *reinterpret_cast<uint16_t*>(&b);
  }
}
__global__ void diff_bf16(const uint16_t* a, const uint16_t* b, size_t n, float* out) {  // out[0] = max |a-b|, out[1] = max |a|
  float md = 0.f, ma = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float x = __uint_as_float((uint32_t)a[i] << 16), y = __uint_as_float((uint32_t)b[i] << 16);
    md = fmaxf(md, fabsf(x - y));
    ma = fmaxf(ma, fabsf(x));
    if (!(x == x) || !(y == y)) md = 1e30f;
  }
  atomicMax(reinterpret_cast<int*>(out), __float_as_int(md));
  atomicMax(reinterpret_cast<int*>(out) + 1, __float_as_int(ma));
}
__global__ void diff_bf16_f32(const uint16_t* a, const float* b, size_t n, float* out) {
  float md = 0.f, ma = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float x = __uint_as_float((uint32_t)a[i] << 16), y = b[i];
    md = fmaxf(md, fabsf(x - y));
    ma = fmaxf(ma, fabsf(y));
    if (!(x == x)) md = 1e30f;
  }
  atomicMax(reinterpret_cast<int*>(out), __float_as_int(md));
  atomicMax(reinterpret_cast<int*>(out) + 1, __float_as_int(ma));
}

__global__ void diff_f32(const float* a, const float* b, size_t n, float* out) {
  float md = 0.f, ma = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    md = fmaxf(md, fabsf(a[i] - b[i]));
    ma = fmaxf(ma, fabsf(b[i]));
    if (!(a[i] == a[i])) md = 1e30f;
  }
  atomicMax(reinterpret_cast<int*>(out), __float_as_int(md));
  atomicMax(reinterpret_cast<int*>(out) + 1, __float_as_int(ma));
}

struct Layer { const char* name; int N, H, W, Ci, Co, k, s, p; int kw = 0; };  // kw = 1: taps along H only

int main(int argc, char** argv) {
  const int F = getenv("NTB_FRAMES") ? atoi(getenv("NTB_FRAMES")) : 512;
  const int reps = getenv("NTB_REPS") ? atoi(getenv("NTB_REPS")) : 5;
  const bool naive = getenv("NTB_NAIVE") != nullptr;
  std::vector<Layer> all = {
      {"l1b0.c1", F, 56, 56, 64, 64, 1, 1, 0},     {"l1b0.c2", F, 56, 56, 64, 64, 3, 1, 1},     {"l1b0.c3", F, 56, 56, 64, 256, 1, 1, 0},
      {"l1b1.c1", F, 56, 56, 256, 64, 1, 1, 0},    {"l2b0.c1", F, 56, 56, 256, 128, 1, 1, 0},
      {"l2b0.c3", F, 28, 28, 128, 512, 1, 1, 0},   {"l2b1.c1", F, 28, 28, 512, 128, 1, 1, 0},
      {"l2b1.c2", F, 28, 28, 128, 128, 3, 1, 1},   {"l3b0.c1", F, 28, 28, 512, 256, 1, 1, 0},
      {"l3b0.c2", F, 28, 28, 256, 256, 3, 2, 1},   {"l3b0.c3", F, 14, 14, 256, 1024, 1, 1, 0},
      {"l3b0.ds", F, 28, 28, 512, 1024, 1, 2, 0},  {"l3b1.c1", F, 14, 14, 1024, 256, 1, 1, 0},
      {"l3b1.c2", F, 14, 14, 256, 256, 3, 1, 1},   {"l4b0.c1", F, 14, 14, 1024, 512, 1, 1, 0},
      {"l4b0.c2", F, 14, 14, 512, 512, 3, 2, 1},   {"l4b0.c3", F, 7, 7, 512, 2048, 1, 1, 0},
      {"l4b0.ds", F, 14, 14, 1024, 2048, 1, 2, 0}, {"l4b1.c1", F, 7, 7, 2048, 512, 1, 1, 0},
      {"l4b1.c2", F, 7, 7, 512, 512, 3, 1, 1},     {"gemm4k", 1, 1, 4096, 4096, 4096, 1, 1, 0},
      {"gemm8k", 1, 1, 8192, 8192, 8192, 1, 1, 0}, {"odd", 3, 13, 11, 192, 320, 3, 1, 1},
      {"odd1", 5, 9, 7, 136, 200, 1, 1, 0},
      // Swin-T linear layers at 8 clips x 16 frames (tokens per frame 3136 / 784 / 196 / 49): qkv, proj, fc1, fc2
      {"sw1.qkv", 128, 56, 56, 96, 288, 1, 1, 0},   {"sw1.proj", 128, 56, 56, 96, 96, 1, 1, 0},
      {"sw1.fc1", 128, 56, 56, 96, 384, 1, 1, 0},   {"sw1.fc2", 128, 56, 56, 384, 96, 1, 1, 0},
      {"sw2.qkv", 128, 28, 28, 192, 576, 1, 1, 0},  {"sw2.proj", 128, 28, 28, 192, 192, 1, 1, 0},
      {"sw2.fc1", 128, 28, 28, 192, 768, 1, 1, 0},  {"sw2.fc2", 128, 28, 28, 768, 192, 1, 1, 0},
      {"sw3.qkv", 128, 14, 14, 384, 1152, 1, 1, 0}, {"sw3.proj", 128, 14, 14, 384, 384, 1, 1, 0},
      {"sw3.fc1", 128, 14, 14, 384, 1536, 1, 1, 0}, {"sw3.fc2", 128, 14, 14, 1536, 384, 1, 1, 0},
      {"sw4.qkv", 128, 7, 7, 768, 2304, 1, 1, 0},   {"sw4.proj", 128, 7, 7, 768, 768, 1, 1, 0},
      {"sw4.fc1", 128, 7, 7, 768, 3072, 1, 1, 0},   {"sw4.fc2", 128, 7, 7, 3072, 768, 1, 1, 0},
      // CRNN / decoder linear layers at 512 rows
      {"fc1.crnn", 512, 1, 1, 36864, 1024, 1, 1, 0}, {"fc2.crnn", 512, 1, 1, 1024, 768, 1, 1, 0},
      {"lstm.ih", 512, 1, 1, 512, 2048, 1, 1, 0},
      // ResNet stem over row triples: 3 taps along H of 64 channels (3 rows x 7 taps x 3 colours)
      {"stem.rows", F, 112, 112, 64, 64, 3, 1, 1, 1},
      // ResNet stem as it runs: width-unfolded, 7 taps along H (stride 2) of 24 channels
      {"stem.unf", F, 224, 112, 24, 64, 7, 2, 3, 1},
  };
  std::vector<Layer> sel;
  for (auto& l : all) {
    if (argc < 2) { if (l.name[0] == 'l' && l.name[1] != 's') sel.push_back(l); continue; }
    for (int i = 1; i < argc; ++i) if (strstr(l.name, argv[i])) { sel.push_back(l); break; }
  }
  uint16_t* flush; const size_t flush_bytes = 1ull << 30;
  HIP(hipMalloc(&flush, flush_bytes));
  float* dstat; HIP(hipMalloc(&dstat, 8));
  hipEvent_t e0, e1; HIP(hipEventCreate(&e0)); HIP(hipEventCreate(&e1));
  printf("%-10s %-28s %8s | %-7s %8s %8s | %8s %8s | %9s %9s\n", "layer", "N,H,W,Ci->Co k s", "GF", "pass", "nt8 ms", "nt8 TF", "old ms", "old TF", "d(new,old)", "d(naive)");
  for (auto& l : sel) {
    const int kw = l.kw ? l.kw : l.k, pw = l.kw ? 0 : l.p, sw = l.kw ? 1 : l.s;
    const int Ho = (l.H + 2 * l.p - l.k) / l.s + 1, Wo = (l.W + 2 * pw - kw) / sw + 1, taps = l.k * kw;
    const int desc[18] = {l.N, 1, l.H, l.W, l.Ci, 1, Ho, Wo, l.Co, 1, l.k, kw, 1, l.s, sw, 0, l.p, pw};
    const size_t nx = (size_t)l.N * l.H * l.W * l.Ci, ny = (size_t)l.N * Ho * Wo * l.Co, nw = (size_t)l.Co * taps * l.Ci;
    uint16_t *x, *dy, *wf, *wt, *y0, *y1, *dx0, *dx1; float* yn;
    HIP(hipMalloc(&x, nx * 2)); HIP(hipMalloc(&dy, ny * 2)); HIP(hipMalloc(&wf, nw * 2)); HIP(hipMalloc(&wt, nw * 2));
    HIP(hipMalloc(&y0, ny * 2)); HIP(hipMalloc(&y1, ny * 2)); HIP(hipMalloc(&dx0, nx * 2)); HIP(hipMalloc(&dx1, nx * 2));
    HIP(hipMalloc(&yn, (nx > ny ? nx : ny) * 4));
    fill_bf16<<<2048, 256>>>(x, nx, 1u, 1.0f); fill_bf16<<<2048, 256>>>(dy, ny, 2u, 1.0f);
    fill_bf16<<<2048, 256>>>(wf, nw, 3u, 0.05f); fill_bf16<<<2048, 256>>>(wt, nw, 4u, 0.05f);
    const double gf = 2.0 * l.N * Ho * Wo * (double)l.Co * l.Ci * taps / 1e9;
    char shape[64]; snprintf(shape, sizeof shape, "%d,%d,%d,%d->%d k%d s%d", l.N, l.H, l.W, l.Ci, l.Co, l.k, l.s);
    for (int pass = 0; pass < 2; ++pass) {
      if (getenv("NTB_PASS") && atoi(getenv("NTB_PASS")) != pass) continue;
      float ms[2] = {0, 0};
      for (int variant = 0; variant < 2; ++variant) {
        setenv("VC_NT8", variant == 0 ? "1" : "0", 1);
        uint16_t* o = pass == 0 ? (variant == 0 ? y0 : y1) : (variant == 0 ? dx0 : dx1);
        HIP(hipMemset(o, getenv("NTB_BETA") ? 0 : 0xFF, (pass == 0 ? ny : nx) * 2));
        auto run = [&]() {
          int rc = pass == 0 ? vc_conv_fwd(desc, 1, x, 0, wf, nullptr, o, 0, 0, 0, 0, nullptr)
                             : vc_conv_dgrad(desc, 1, dy, 0, wt, o, 0, 0, getenv("NTB_BETA") ? 1 : 0, nullptr);
          if (rc) { printf("%s failed: %s\n", l.name, vc_last_error()); exit(3); }
        };
        run();
        HIP(hipDeviceSynchronize());
        float tot = 0;
        for (int r = 0; r < reps; ++r) {
          if (!getenv("NTB_NOFLUSH")) HIP(hipMemsetAsync(flush, r, flush_bytes, nullptr));
          HIP(hipEventRecord(e0, nullptr));
          run();
          HIP(hipEventRecord(e1, nullptr));
          HIP(hipEventSynchronize(e1));
          float t; HIP(hipEventElapsedTime(&t, e0, e1)); tot += t;
        }
        ms[variant] = tot / reps;
      }
      const size_t no = pass == 0 ? ny : nx;
      HIP(hipMemset(dstat, 0, 8));
      diff_bf16<<<1024, 256>>>(pass == 0 ? y0 : dx0, pass == 0 ? y1 : dx1, no, dstat);
      float d[2]; HIP(hipMemcpy(d, dstat, 8, hipMemcpyDeviceToHost));
      float dn[2] = {0, 1};
      if (naive) {
        int rc = vc_conv_fwd_naive_bf16(desc, pass, pass == 0 ? x : dy, pass == 0 ? wf : wt, nullptr, yn, 0, 1, nullptr);
        if (rc) { printf("naive failed: %s\n", vc_last_error()); exit(3); }
        HIP(hipMemset(dstat, 0, 8));
        diff_bf16_f32<<<1024, 256>>>(pass == 0 ? y0 : dx0, yn, no, dstat);
        HIP(hipMemcpy(dn, dstat, 8, hipMemcpyDeviceToHost));
      }
      printf("%-10s %-28s %8.1f | %-7s %8.3f %8.0f | %8.3f %8.0f | %9.2e %9.2e\n", l.name, shape, gf, pass == 0 ? "fwd" : "dgrad",
             ms[0], gf / ms[0], ms[1], gf / ms[1], d[0] / (d[1] > 0 ? d[1] : 1), dn[0] / (dn[1] > 0 ? dn[1] : 1));
      fflush(stdout);
    }
    if (!getenv("NTB_PASS") || atoi(getenv("NTB_PASS")) == 2) {  // weight gradient: 256-wide tiles against 128-wide
      float *dw0, *dw1, *dwn, *db = nullptr; float ms[2] = {0, 0};
      HIP(hipMalloc(&dw0, nw * 4)); HIP(hipMalloc(&dw1, nw * 4)); HIP(hipMalloc(&dwn, nw * 4));
      if (getenv("NTB_BIAS")) { HIP(hipMalloc(&db, l.Co * 4)); HIP(hipMemset(db, 0, l.Co * 4)); }  // with the bias gradient
      for (int variant = 0; variant < 2; ++variant) {
        setenv("VC_WGRAD8", variant == 0 ? "1" : "0", 1);
        float* o = variant == 0 ? dw0 : dw1;
        auto run = [&]() {
          int rc = vc_conv_wgrad(desc, 1, x, 0, dy, 0, o, db, 0, nullptr);
          if (rc) { printf("%s wgrad failed: %s\n", l.name, vc_last_error()); exit(3); }
        };
        run();
        HIP(hipDeviceSynchronize());
        float tot = 0;
        for (int r = 0; r < reps; ++r) {
          if (!getenv("NTB_NOFLUSH")) HIP(hipMemsetAsync(flush, r, flush_bytes, nullptr));
          if (r == reps - 1) HIP(hipMemsetAsync(o, 0, nw * 4, nullptr));  // the last launch's sum is the one compared
          HIP(hipEventRecord(e0, nullptr));
          run();
          HIP(hipEventRecord(e1, nullptr));
          HIP(hipEventSynchronize(e1));
          float t; HIP(hipEventElapsedTime(&t, e0, e1)); tot += t;
        }
        ms[variant] = tot / reps;
      }
      unsetenv("VC_WGRAD8");
      HIP(hipMemset(dstat, 0, 8));
      diff_f32<<<1024, 256>>>(dw0, dw1, nw, dstat);
      float d[2]; HIP(hipMemcpy(d, dstat, 8, hipMemcpyDeviceToHost));
      float dn[2] = {0, 1};
      if (naive) {
        HIP(hipMemset(dwn, 0, nw * 4));
        int rc = vc_conv_wgrad(desc, 1, x, 0, dy, 0, dwn, nullptr, 1, nullptr);
        if (rc) { printf("naive wgrad failed: %s\n", vc_last_error()); exit(3); }
        HIP(hipMemset(dstat, 0, 8));
        diff_f32<<<1024, 256>>>(dw0, dwn, nw, dstat);
        HIP(hipMemcpy(dn, dstat, 8, hipMemcpyDeviceToHost));
      }
      printf("%-10s %-28s %8.1f | %-7s %8.3f %8.0f | %8.3f %8.0f | %9.2e %9.2e\n", l.name, shape, gf, "wgrad", ms[0], gf / ms[0],
             ms[1], gf / ms[1], d[0] / (d[1] > 0 ? d[1] : 1), dn[0] / (dn[1] > 0 ? dn[1] : 1));
      fflush(stdout);
      (void)hipFree(dw0); (void)hipFree(dw1); (void)hipFree(dwn); if (db) (void)hipFree(db);
    }
    (void)hipFree(x); (void)hipFree(dy); (void)hipFree(wf); (void)hipFree(wt); (void)hipFree(y0); (void)hipFree(y1); (void)hipFree(dx0); (void)hipFree(dx1); (void)hipFree(yn);
  }
  return 0;
}
