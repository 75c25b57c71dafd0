// valu_rate.hip -- how many cycles does a SIMD of gfx950 spend per wave64 integer VALU instruction, and do two or more
// waves on one SIMD overlap theirs?  (MI355X_MICROARCH.md quotes 2 cycles for v_fma_f32 with several waves, 4 for a lone
// wave; the fill kernels issue v_add_u32 / v_max_i32 / v_max3_i32 almost exclusively, and their counters show exactly one
// quad-cycle of SQ_ACTIVE_INST_VALU per instruction.)  Each wave runs a long unrolled stream of independent instructions
// on 8 accumulators and stamps s_memtime around it; the grid puts 1, 2, 3 or 4 waves on every SIMD.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rate tools/valu_rate.hip && /tmp/valu_rate
// Prints, per instruction kind and waves/SIMD: cycles per instruction as one wave sees them, and SIMD cycles per
// wave-instruction (= the former / waves per SIMD): the second column is what bounds a VALU-bound kernel.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int UNROLL = 64, ITER = 4096;

#define OPS8(OP)                                                                              \
  OP(a0, b0) OP(a1, b1) OP(a2, b2) OP(a3, b3) OP(a4, b0) OP(a5, b1) OP(a6, b2) OP(a7, b3)

#define ADD_U32(a, b) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
#define MAX_I32(a, b) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a) : "v"(b));
#define MAX3_I32(a, b) asm volatile("v_max3_i32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
#define PK_ADD_I16(a, b) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a) : "v"(b));
#define PK_MAX_I16(a, b) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a) : "v"(b));
#define ADD_F32(a, b) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
#define MOV_B32(a, b) asm volatile("v_mov_b32 %0, %1" : "+v"(a) : "v"(b));
#define CNDMASK(a, b) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b));
#define PERM_B32(a, b) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
#define OR3_B32(a, b) asm volatile("v_or3_b32 %0, %0, %1, %1" : "+v"(a) : "v"(b));

#define KERNEL(NAME, OP)                                                                         \
  __global__ void __launch_bounds__(256) NAME(int* out, long long* cyc, int seed) {              \
    int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    int b0 = seed, b1 = seed + 1, b2 = seed + 2, b3 = seed + 3;                                  \
    const long long t0 = __builtin_amdgcn_s_memtime();                                           \
    for (int it = 0; it < ITER; ++it) {                                                          \
      OPS8(OP) OPS8(OP) OPS8(OP) OPS8(OP) OPS8(OP) OPS8(OP) OPS8(OP) OPS8(OP)                    \
    }                                                                                            \
    const long long t1 = __builtin_amdgcn_s_memtime();                                           \
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                 \
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;             \
  }

KERNEL(k_add_u32, ADD_U32)
KERNEL(k_max_i32, MAX_I32)
KERNEL(k_max3_i32, MAX3_I32)
KERNEL(k_pk_add_i16, PK_ADD_I16)
KERNEL(k_pk_max_i16, PK_MAX_I16)
KERNEL(k_add_f32, ADD_F32)
KERNEL(k_mov_b32, MOV_B32)
KERNEL(k_cndmask, CNDMASK)
KERNEL(k_perm_b32, PERM_B32)
KERNEL(k_or3_b32, OR3_B32)

typedef void (*kern_t)(int*, long long*, int);

int main() {
  int dev = 0;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, dev));
  const int cus = prop.multiProcessorCount;
  int* out;
  long long* cyc;
  CK(hipMalloc(&out, sizeof(int) * 256 * cus * 8));
  CK(hipMalloc(&cyc, sizeof(long long) * 4 * cus * 8));
  struct { const char* name; kern_t k; } kinds[] = {
      {"v_add_u32", k_add_u32}, {"v_max_i32", k_max_i32}, {"v_max3_i32", k_max3_i32}, {"v_pk_add_i16", k_pk_add_i16},
      {"v_pk_max_i16", k_pk_max_i16}, {"v_add_f32", k_add_f32}, {"v_mov_b32", k_mov_b32}, {"v_cndmask_b32", k_cndmask},
      {"v_perm_b32", k_perm_b32}, {"v_or3_b32", k_or3_b32}};
  printf("%-14s %10s %22s %26s\n", "instruction", "waves/SIMD", "cycles/instr (a wave)", "SIMD cycles/wave-instr");
  for (auto& kd : kinds) {
    for (int wps = 1; wps <= 4; ++wps) {
      const int blocks = cus * wps;  // 256-thread blocks: one wave per SIMD each; wps blocks per CU
      for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kd.k, dim3(blocks), dim3(256), 0, 0, out, cyc, rep);
      CK(hipDeviceSynchronize());
      std::vector<long long> h(4 * blocks);
      CK(hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
      std::sort(h.begin(), h.end());
      const double med = (double)h[h.size() / 2] / ((double)UNROLL * ITER);
      printf("%-14s %10d %22.2f %26.2f\n", kd.name, wps, med, med / wps);
    }
  }
  return 0;
}
