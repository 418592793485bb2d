// gfx950 integer / fp64 issue-rate microbenchmark.
// Decides the limb representation for the Fr/Fp Montgomery multiplier.
// Each kernel runs ITER iterations of 8 independent dependency chains per lane;
// reported: wave-instructions per cycle per SIMD assuming 2.4 GHz is NOT assumed -
// we report ns per wave-instruction per SIMD and Gops/s chip-wide instead.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITER = 4096;
constexpr int CH = 8;

struct OpMad64 {  // v_mad_u64_u32
  static constexpr const char* name = "v_mad_u64_u32";
  __device__ static void run(uint64_t (&a)[CH], uint32_t x, uint32_t y) {
#pragma unroll
    for (int c = 0; c < CH; ++c) a[c] = (uint64_t)(uint32_t)a[c] * (uint64_t)x + (a[c] >> 7) + y;
  }
};
struct OpMad64Pure {  // v_mad_u64_u32 only, via asm
  static constexpr const char* name = "v_mad_u64_u32(asm)";
  __device__ static void run(uint64_t (&a)[CH], uint32_t x, uint32_t y) {
#pragma unroll
    for (int c = 0; c < CH; ++c)
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[c]) : "v"(x), "v"(y) : "vcc");
  }
};
struct OpMulLo {
  static constexpr const char* name = "v_mul_lo_u32(asm)";
  __device__ static void run(uint64_t (&a)[CH], uint32_t x, uint32_t y) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      uint32_t t = (uint32_t)a[c];
      asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(t) : "v"(x));
      a[c] = t;
    }
  }
};
struct OpMulHi {
  static constexpr const char* name = "v_mul_hi_u32(asm)";
  __device__ static void run(uint64_t (&a)[CH], uint32_t x, uint32_t y) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      uint32_t t = (uint32_t)a[c];
      asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(t) : "v"(x));
      a[c] = t;
    }
  }
};
struct OpMad24 {
  static constexpr const char* name = "v_mad_u32_u24(asm)";
  __device__ static void run(uint64_t (&a)[CH], uint32_t x, uint32_t y) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      uint32_t t = (uint32_t)a[c];
      asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(t) : "v"(x), "v"(y));
      a[c] = t;
    }
  }
};
struct OpMulHi24 {
  static constexpr const char* name = "v_mul_hi_u32_u24(asm)";
  __device__ static void run(uint64_t (&a)[CH], uint32_t x, uint32_t y) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      uint32_t t = (uint32_t)a[c];
      asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(t) : "v"(x));
      a[c] = t;
    }
  }
};
struct OpAddCo {
  static constexpr const char* name = "v_add_co_u32+v_addc_co_u32 pair(asm)";
  __device__ static void run(uint64_t (&a)[CH], uint32_t x, uint32_t y) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      uint32_t lo = (uint32_t)a[c], hi = (uint32_t)(a[c] >> 32);
      asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo), "+v"(hi) : "v"(x), "v"(y) : "vcc");
      a[c] = ((uint64_t)hi << 32) | lo;
    }
  }
};
struct OpAdd3 {
  static constexpr const char* name = "v_add3_u32(asm)";
  __device__ static void run(uint64_t (&a)[CH], uint32_t x, uint32_t y) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      uint32_t t = (uint32_t)a[c];
      asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(t) : "v"(x), "v"(y));
      a[c] = t;
    }
  }
};
struct OpFma64 {
  static constexpr const char* name = "v_fma_f64(asm)";
  __device__ static void run(uint64_t (&a)[CH], uint32_t x, uint32_t y) {
    double dx = __hiloint2double(0x3ff00000, x), dy = __hiloint2double(0x3ff00000, y);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      double t = __longlong_as_double(a[c]);
      asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(t) : "v"(dx), "v"(dy));
      a[c] = __double_as_longlong(t);
    }
  }
};
struct OpMul64f {
  static constexpr const char* name = "v_mul_f64(asm)";
  __device__ static void run(uint64_t (&a)[CH], uint32_t x, uint32_t y) {
    double dx = __hiloint2double(0x3ff00000, x);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      double t = __longlong_as_double(a[c]);
      asm volatile("v_mul_f64 %0, %0, %1" : "+v"(t) : "v"(dx));
      a[c] = __double_as_longlong(t);
    }
  }
};
struct OpFma32 {
  static constexpr const char* name = "v_fma_f32(asm)";
  __device__ static void run(uint64_t (&a)[CH], uint32_t x, uint32_t y) {
    float fx = __uint_as_float(0x3f800000 | (x & 0xffff)), fy = __uint_as_float(y);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      float t = __uint_as_float((uint32_t)a[c]);
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(t) : "v"(fx), "v"(fy));
      a[c] = __float_as_uint(t);
    }
  }
};
struct OpMadI64I32 {
  static constexpr const char* name = "v_mad_i64_i32(asm)";
  __device__ static void run(uint64_t (&a)[CH], uint32_t x, uint32_t y) {
#pragma unroll
    for (int c = 0; c < CH; ++c)
      asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(a[c]) : "v"(x), "v"(y) : "vcc");
  }
};
struct OpLshlAdd64 {
  static constexpr const char* name = "v_lshl_add_u64(asm)";
  __device__ static void run(uint64_t (&a)[CH], uint32_t x, uint32_t y) {
    uint64_t b = ((uint64_t)x << 32) | y;
#pragma unroll
    for (int c = 0; c < CH; ++c)
      asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[c]) : "v"(b));
  }
};


// ---- round 3: the simple 32-bit integer ops the field code's bookkeeping is made of ----
#define SIMPLE_OP(NAME, TEXT, ASM)                                                       \
  struct NAME {                                                                          \
    static constexpr const char* name = TEXT;                                            \
    __device__ static void run(uint64_t (&a)[CH], uint32_t x, uint32_t y) {              \
      _Pragma("unroll") for (int c = 0; c < CH; ++c) {                                   \
        uint32_t t = (uint32_t)a[c];                                                     \
        asm volatile(ASM : "+v"(t) : "v"(x), "v"(y));                                    \
        a[c] = t;                                                                        \
      }                                                                                  \
    }                                                                                    \
  };
SIMPLE_OP(OpAddU32, "v_add_u32(asm)", "v_add_u32 %0, %0, %1")
SIMPLE_OP(OpSubU32, "v_sub_u32(asm)", "v_sub_u32 %0, %0, %1")
SIMPLE_OP(OpAndB32, "v_and_b32(asm)", "v_and_b32 %0, %0, %1")
SIMPLE_OP(OpOrB32, "v_or_b32(asm)", "v_or_b32 %0, %0, %1")
SIMPLE_OP(OpXorB32, "v_xor_b32(asm)", "v_xor_b32 %0, %0, %1")
SIMPLE_OP(OpLshr, "v_lshrrev_b32(asm)", "v_lshrrev_b32 %0, 3, %0")
SIMPLE_OP(OpLshl, "v_lshlrev_b32(asm)", "v_lshlrev_b32 %0, 3, %0")
SIMPLE_OP(OpAshr, "v_ashrrev_i32(asm)", "v_ashrrev_i32 %0, 3, %0")
SIMPLE_OP(OpMov, "v_mov_b32(asm)", "v_mov_b32 %0, %1")
SIMPLE_OP(OpBfe, "v_bfe_u32(asm)", "v_bfe_u32 %0, %0, 3, 29")
SIMPLE_OP(OpAlignbit, "v_alignbit_b32(asm)", "v_alignbit_b32 %0, %0, %1, 29")
SIMPLE_OP(OpAndOr, "v_and_or_b32(asm)", "v_and_or_b32 %0, %0, %1, %2")
SIMPLE_OP(OpLshlAdd32, "v_lshl_add_u32(asm)", "v_lshl_add_u32 %0, %0, 3, %1")
SIMPLE_OP(OpAddLshl, "v_add_lshl_u32(asm)", "v_add_lshl_u32 %0, %0, %1, 3")
SIMPLE_OP(OpLshlOr, "v_lshl_or_b32(asm)", "v_lshl_or_b32 %0, %0, 3, %1")
SIMPLE_OP(OpCndmask, "v_cndmask_b32(asm)", "v_cndmask_b32 %0, %0, %1, vcc")
SIMPLE_OP(OpMulU24, "v_mul_u32_u24(asm)", "v_mul_u32_u24 %0, %0, %1")
SIMPLE_OP(OpMaxU32, "v_max_u32(asm)", "v_max_u32 %0, %0, %1")
SIMPLE_OP(OpPkAddU16, "v_pk_add_u16(asm)", "v_pk_add_u16 %0, %0, %1")
SIMPLE_OP(OpAddF32, "v_add_f32(asm)", "v_add_f32 %0, %0, %1")
SIMPLE_OP(OpAddCoOnly, "v_add_co_u32(asm)", "v_add_co_u32 %0, vcc, %0, %1")
SIMPLE_OP(OpSubrevCo, "v_subb_co_u32(asm)", "v_subb_co_u32 %0, vcc, %0, %1, vcc")
SIMPLE_OP(OpPerm, "v_perm_b32(asm)", "v_perm_b32 %0, %0, %1, %2")
SIMPLE_OP(OpDot4, "v_dot4_u32_u8(asm)", "v_dot4_u32_u8 %0, %0, %1, %2")
#define WIDE_OP(NAME, TEXT, ASM)                                                         \
  struct NAME {                                                                          \
    static constexpr const char* name = TEXT;                                            \
    __device__ static void run(uint64_t (&a)[CH], uint32_t x, uint32_t y) {              \
      _Pragma("unroll") for (int c = 0; c < CH; ++c) asm volatile(ASM : "+v"(a[c]) : "v"(x), "v"(y)); \
    }                                                                                    \
  };
WIDE_OP(OpLshr64, "v_lshrrev_b64(asm)", "v_lshrrev_b64 %0, 29, %0")
WIDE_OP(OpLshl64, "v_lshlrev_b64(asm)", "v_lshlrev_b64 %0, 3, %0")
WIDE_OP(OpAshr64, "v_ashrrev_i64(asm)", "v_ashrrev_i64 %0, 29, %0")
WIDE_OP(OpMov64, "v_mov_b64(asm)", "v_mov_b64 %0, %0")
SIMPLE_OP(OpLshlV, "v_lshlrev_b32 by vgpr(asm)", "v_lshlrev_b32 %0, %1, %0")
SIMPLE_OP(OpLshrV, "v_lshrrev_b32 by vgpr(asm)", "v_lshrrev_b32 %0, %1, %0")
SIMPLE_OP(OpLshl16, "v_lshlrev_b32 by 16(asm)", "v_lshlrev_b32 %0, 16, %0")
SIMPLE_OP(OpSubrev, "v_subrev_u32(asm)", "v_subrev_u32 %0, %0, %1")
SIMPLE_OP(OpAddDpp, "v_add_u32 dpp quad_perm(asm)", "v_add_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
SIMPLE_OP(OpMovDpp, "v_mov_b32 dpp row_shr:1(asm)", "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
SIMPLE_OP(OpMadU16, "v_mad_u32_u16(asm)", "v_mad_u32_u16 %0, %0, %1, %2")

template <class Op>
__global__ __launch_bounds__(256) void k(uint64_t* out, uint32_t x, uint32_t y) {
  uint64_t a[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) a[c] = threadIdx.x * 977u + c * 131u + blockIdx.x;
  for (int it = 0; it < ITER; ++it) Op::run(a, x, y);
  uint64_t s = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) s ^= a[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class Op>
int bench(uint64_t* d_out, int blocks, double opsPerRun) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  std::vector<float> ts;
  for (int r = 0; r < 7; ++r) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<Op>, dim3(blocks), dim3(256), 0, 0, d_out, 0x9e3779b9u + r, 0x7f4a7c15u);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  double ms = ts[ts.size() / 2];
  // wave-instructions issued in total
  double waveInstr = (double)blocks * 4 /*waves per block*/ * ITER * CH * opsPerRun;
  double perSimd = waveInstr / 1024.0;  // 256 CUs x 4 SIMD
  double nsPer = ms * 1e6 / perSimd;
  printf("%-40s blocks=%5d  %.3f ms  %.3f ns/wave-instr/SIMD  (= %.2f cyc @2.4GHz)  %.1f Gop/s lane-ops chip\n",
         Op::name, blocks, ms, nsPer, nsPer * 2.4, waveInstr * 64 / (ms * 1e-3) / 1e9);
  return 0;
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs=%d clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  int blocks = 256 * 8;  // 8 blocks/CU = 32 waves/CU = 8 waves/SIMD
  uint64_t* d_out; CHECK(hipMalloc(&d_out, (size_t)blocks * 256 * 8));
  for (int rep = 0; rep < 2; ++rep) {
    if (bench<OpFma32>(d_out, blocks, 1)) return 1;
    if (bench<OpMad64>(d_out, blocks, 1)) return 1;
    if (bench<OpMad64Pure>(d_out, blocks, 1)) return 1;
    if (bench<OpMadI64I32>(d_out, blocks, 1)) return 1;
    if (bench<OpMulLo>(d_out, blocks, 1)) return 1;
    if (bench<OpMulHi>(d_out, blocks, 1)) return 1;
    if (bench<OpMad24>(d_out, blocks, 1)) return 1;
    if (bench<OpMulHi24>(d_out, blocks, 1)) return 1;
    if (bench<OpAddCo>(d_out, blocks, 2)) return 1;
    if (bench<OpAdd3>(d_out, blocks, 1)) return 1;
    if (bench<OpLshlAdd64>(d_out, blocks, 1)) return 1;
    if (bench<OpFma64>(d_out, blocks, 1)) return 1;
    if (bench<OpMul64f>(d_out, blocks, 1)) return 1;
  }
  printf("--- round 3: simple 32-bit integer ops, 8 waves/SIMD ---\n");
#define RUN(OP) if (bench<OP>(d_out, blocks, 1)) return 1;
  RUN(OpFma32) RUN(OpAddF32) RUN(OpAddU32) RUN(OpSubU32) RUN(OpAndB32) RUN(OpOrB32) RUN(OpXorB32) RUN(OpLshr) RUN(OpLshl)
  RUN(OpAshr) RUN(OpMov) RUN(OpBfe) RUN(OpAlignbit) RUN(OpAndOr) RUN(OpLshlAdd32) RUN(OpAddLshl) RUN(OpLshlOr)
  RUN(OpCndmask) RUN(OpMulU24) RUN(OpMaxU32) RUN(OpPkAddU16) RUN(OpAddCoOnly) RUN(OpSubrevCo) RUN(OpPerm) RUN(OpDot4)
  RUN(OpMadU16) RUN(OpAdd3) RUN(OpMad64Pure)
  RUN(OpLshr64) RUN(OpLshl64) RUN(OpAshr64) RUN(OpMov64) RUN(OpLshlV) RUN(OpLshrV) RUN(OpLshl16) RUN(OpSubrev)
  RUN(OpAddDpp) RUN(OpMovDpp)
  printf("--- 4 waves/SIMD ---\n");
  blocks = 256 * 4;
  RUN(OpFma32) RUN(OpAddU32) RUN(OpAndB32) RUN(OpLshr) RUN(OpAdd3) RUN(OpMad64Pure)
  printf("--- 2 waves/SIMD ---\n");
  blocks = 256 * 2;
  RUN(OpFma32) RUN(OpAddU32) RUN(OpAndB32) RUN(OpLshr) RUN(OpAdd3) RUN(OpMad64Pure)
  // one wave per SIMD variant (latency-exposed)
  printf("--- 1 block/CU (1 wave/SIMD) ---\n");
  if (bench<OpMad64Pure>(d_out, 256, 1)) return 1;
  if (bench<OpFma64>(d_out, 256, 1)) return 1;
  if (bench<OpMad24>(d_out, 256, 1)) return 1;
  return 0;
}
