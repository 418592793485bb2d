// Does a DEPENDENT chain of v_mad_u64_u32 run at full rate with W waves per SIMD?
// Decides whether csrc/field.h may serialise a Montgomery column into one multiply-add chain
// (fewer instructions) or needs two interleaved chains per lane (what hipcc emits on its own).
//   hipcc -O3 --offload-arch=gfx950 mad_chain.hip -o mad_chain && ./mad_chain
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

constexpr int ITER = 2048;

template <int CH>
__global__ __launch_bounds__(64) void chain_kernel(uint64_t* out, uint32_t x, uint32_t y) {
  uint64_t a[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) a[c] = threadIdx.x + c;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int rep = 0; rep < 16; ++rep) {
#pragma unroll
      for (int c = 0; c < CH; ++c) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[c]) : "v"(x), "v"(y) : "vcc");
    }
  }
  uint64_t s = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += a[c];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

// the shape of one Montgomery column: chain of mads, then mul_lo / and / mad / shift, all dependent
__global__ __launch_bounds__(64) void column_kernel(uint64_t* out, uint32_t x, uint32_t y) {
  uint64_t acc = threadIdx.x;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int rep = 0; rep < 12; ++rep) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y) : "vcc");
    uint32_t m;
    asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(m) : "v"((uint32_t)acc), "v"(y));
    asm volatile("v_and_b32 %0, 0x3fffffff, %0" : "+v"(m));
    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(m), "v"(x) : "vcc");
    asm volatile("v_lshrrev_b64 %0, 30, %0" : "+v"(acc));
  }
  out[blockIdx.x * 64 + threadIdx.x] = acc;
}

template <class K>
static void run(const char* name, K kern, int waves_per_simd, double instr_per_thread) {
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int blocks = cus * 4 * waves_per_simd;
  uint64_t* d;
  hipMalloc(&d, (size_t)blocks * 64 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, d, 12345u, 678u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, d, 12345u, 678u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  // wave-instructions per SIMD = waves_per_simd * instr_per_thread
  const double cyc = ms * 1e-3 * 2.4e9 / (waves_per_simd * instr_per_thread);
  printf("%-34s waves/SIMD=%d  %.3f ms  %.2f cycles per wave-instruction per SIMD\n", name, waves_per_simd, ms, cyc);
  hipFree(d);
}

int main() {
  for (int w : {1, 2, 3, 4, 8}) {
    run("1 dependent chain", chain_kernel<1>, w, (double)ITER * 16);
    run("2 interleaved chains", chain_kernel<2>, w, (double)ITER * 32);
    run("4 interleaved chains", chain_kernel<4>, w, (double)ITER * 64);
    run("Montgomery column (serial)", column_kernel, w, (double)ITER * 16);
  }
  return 0;
}
