// Does the immediate offset of global_load_lds_dwordx4 move the LDS destination as well as the
// global source?  Build: hipcc -O3 --offload-arch=gfx950 glds_offset.hip -o glds_offset
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(64) void k(const uint32_t* src, uint32_t* out) {
  __shared__ __attribute__((aligned(16))) uint32_t stage[4 * 256];
  const uint32_t lane = threadIdx.x;
  for (int i = lane; i < 4 * 256; i += 64) stage[i] = 0xdeadbeefu;
  __syncthreads();
  const uint32_t* p = src + lane * 32;     // 128-byte rows
  // quad 1 of every row, requested with offset 16 and LDS base = stage + 256 words (second image row)
  __builtin_amdgcn_global_load_lds(p, (__attribute__((address_space(3))) uint32_t*)(stage + 256), 16, 16, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 4 * 256; i += 64) out[i] = stage[i];
}
int main() {
  std::vector<uint32_t> h(64 * 32);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)i;
  uint32_t *d_src, *d_out;
  hipMalloc(&d_src, h.size() * 4);
  hipMalloc(&d_out, 1024 * 4);
  hipMemcpy(d_src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_src, d_out);
  std::vector<uint32_t> o(1024);
  hipMemcpy(o.data(), d_out, 1024 * 4, hipMemcpyDeviceToHost);
  // where did lane 0's quad (words 4..7 of row 0) land?
  for (int i = 0; i < 1024; ++i)
    if (o[i] == 4u) { printf("lane 0 quad landed at LDS word %d (256 = base only, 260 = base + offset)\n", i); break; }
  for (int i = 0; i < 1024; ++i)
    if (o[i] == 32u + 4u) { printf("lane 1 quad landed at LDS word %d\n", i); break; }
  return 0;
}
