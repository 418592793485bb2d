// Does a mostly IDLE third wave on a SIMD slow down two busy waves?  (The commit pipeline keeps 2 accumulate waves per
// SIMD and lets prep / reduce workgroups co-reside; tools/microbench/mad_chain.hip showed that THREE busy waves issue
// worse than two.)  busy: 4 workgroups of 2 waves per CU running dependent v_mad_u64_u32 chains.  side: 1 workgroup of
// 4 waves per CU (one per SIMD) that sleeps, chases pointers through memory, or runs multiply-adds itself.
//   hipcc -O3 -w --offload-arch=gfx950 coresident.hip -o coresident && GPU_MAX_HW_QUEUES=8 ./coresident
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

// clk[0..1]: shader-clock ticks (s_memtime) and 100 MHz ticks (s_memrealtime) wave 0 of workgroup 0 spent in the loop
__global__ __launch_bounds__(128) void busy_kernel(uint64_t* out, uint32_t x, uint32_t y, int iters, long long* clk) {
  uint64_t a = threadIdx.x, b = threadIdx.x + 7;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a) : "v"(x), "v"(y) : "vcc");
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(b) : "v"(y), "v"(x) : "vcc");
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
  out[blockIdx.x * 128 + threadIdx.x] = a + b;
}

// mode 0: s_sleep loop; 1: dependent global loads (latency-bound); 2: multiply-adds
__global__ __launch_bounds__(256) void side_kernel(uint64_t* out, const uint32_t* chase, uint32_t n, int mode, long long cycles) {
  const long long t0 = clock64();
  uint64_t a = threadIdx.x;
  uint32_t p = (blockIdx.x * 256 + threadIdx.x) % n;
  while (clock64() - t0 < cycles) {
    if (mode == 0) {
      __builtin_amdgcn_s_sleep(64);
    } else if (mode == 1) {
      p = chase[p];
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a) : "v"(p), "v"(p) : "vcc");
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a + p;
}

int main() {
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  hipStream_t sa, sb;
  hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
  uint64_t *d1, *d2;
  hipMalloc(&d1, (size_t)cus * 4 * 128 * 8);
  hipMalloc(&d2, (size_t)cus * 256 * 8);
  const uint32_t n = 1u << 24;
  std::vector<uint32_t> h(n);
  for (uint32_t i = 0; i < n; ++i) h[i] = (uint32_t)(((uint64_t)i * 2654435761u + 12345u) % n);
  uint32_t* chase;
  hipMalloc(&chase, (size_t)n * 4);
  hipMemcpy(chase, h.data(), (size_t)n * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  long long* clk;
  hipHostMalloc(&clk, 16);
  const int iters = 20000;     // ~ 1.5 ms of busy work
  double mhz = 0;
  auto run = [&](int mode, int side_wgs_per_cu, int side_threads) {
    if (mode >= 0) hipLaunchKernelGGL(side_kernel, dim3(cus * side_wgs_per_cu), dim3(side_threads), 0, sb, d2, chase, n, mode, 40000000LL);
    hipEventRecord(e0, sa);
    hipLaunchKernelGGL(busy_kernel, dim3(cus * 4), dim3(128), 0, sa, d1, 12345u, 678u, iters, clk);
    hipEventRecord(e1, sa);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipDeviceSynchronize();
    mhz = 100.0 * (double)clk[0] / (double)clk[1];
    return ms;
  };
  run(-1, 0, 0);
  const char* names[] = {"sleeping", "pointer-chasing", "multiply-adds"};
  float t = run(-1, 0, 0);
  printf("busy kernel alone (2 waves/SIMD):                      %.3f ms  shader clock %.0f MHz\n", t, mhz);
  for (int mode = 0; mode < 3; ++mode) {
    t = run(mode, 1, 256);
    printf("+ 1 %-16s wave  per SIMD (1 WG x 256 per CU): %.3f ms  shader clock %.0f MHz\n", names[mode], t, mhz);
    t = run(mode, 2, 256);
    printf("+ 2 %-16s waves per SIMD (2 WG x 256 per CU): %.3f ms  shader clock %.0f MHz\n", names[mode], t, mhz);
  }
  t = run(-1, 0, 0);
  printf("busy kernel alone again:                               %.3f ms  shader clock %.0f MHz\n", t, mhz);
  // 20000 x 32 multiply-adds per wave, 2 waves per SIMD
  printf("=> %.2f shader cycles per v_mad_u64_u32 per SIMD (alone)\n", t * 1e-3 * mhz * 1e6 / (20000.0 * 32 * 2));
  return 0;
}
