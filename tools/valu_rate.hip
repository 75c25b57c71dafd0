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
#include <cstring>
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
#define SUB_U32(a, b) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a) : "v"(b));
#define MIN_U32(a, b) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a) : "v"(b));
#define ADD3_U32(a, b) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
#define LSHL_ADD(a, b) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a) : "v"(b));
#define BFI_B32(a, b) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(a) : "v"(b));
#define ASHR_I32(a, b) asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(a) : "v"(b));
#define AND_B32(a, b) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a) : "v"(b));
#define OR_B32(a, b) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a) : "v"(b));
#define AND_OR(a, b) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
#define MED3_I32(a, b) asm volatile("v_med3_i32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
#define MAX_U32(a, b) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a) : "v"(b));
#define MAX_I16(a, b) asm volatile("v_max_i16 %0, %0, %1" : "+v"(a) : "v"(b));
#define MAX_F32(a, b) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(b));
#define CND_E64(a, b) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a) : "v"(b) : "s20", "s21");
#define CMP_E64(a, b) asm volatile("v_cmp_lt_i32_e64 s[20:21], %0, %1" : : "v"(a), "v"(b) : "s20", "s21");
#define CMP_VCC(a, b) asm volatile("v_cmp_lt_i32_e32 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc");
#define CMP_CND(a, b) asm volatile("v_cmp_lt_i32_e64 s[20:21], %0, %1\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a) : "v"(b) : "s20", "s21");
#define DPP_MIN(a, b) asm volatile("v_min_i32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a) : "v"(b));
#define DPP_MOV(a, b) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a) : "v"(b));
#define MOV_B64(a, b) asm volatile("v_mov_b64 %0, %1" : "+v"(a##w) : "v"(b##w));
#define SUBREV_CO(a, b) asm volatile("v_subrev_co_u32 %0, vcc, %0, %1" : "+v"(a) : "v"(b) : "vcc");
#define MAD_I24(a, b) asm volatile("v_mad_i32_i24 %0, %0, %1, %1" : "+v"(a) : "v"(b));
#define ADD_SGPR(a, b) asm volatile("v_add_u32 %0, s20, %0" : "+v"(a) : "v"(b) : "s20");
#define ADD_INL(a, b) asm volatile("v_add_u32 %0, 5, %0" : "+v"(a) : "v"(b));
#define ADD_LIT(a, b) asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(a) : "v"(b));
#define MAX3_SGPR(a, b) asm volatile("v_max3_i32 %0, %0, %1, s20" : "+v"(a) : "v"(b) : "s20");
#define MIX_ADD_MAX(a, b) asm volatile("v_add_u32 %0, %0, %1\n\tv_max_i32 %0, %0, %1" : "+v"(a) : "v"(b));
#define MIX_ADD_MAX3(a, b) asm volatile("v_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_max3_i32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
#define MIX_SGPR(a, b) asm volatile("v_add_u32 %0, s20, %0\n\tv_add_u32 %0, s21, %0\n\tv_add_u32 %0, %0, %1\n\tv_max3_i32 %0, %0, %1, %1" : "+v"(a) : "v"(b) : "s20", "s21");
#define ADD_SALU(a, b) asm volatile("v_add_u32 %0, %0, %1\n\ts_add_u32 s20, s20, 1" : "+v"(a) : "v"(b) : "s20", "scc");
#define MAX3_SALU(a, b) asm volatile("v_max3_i32 %0, %0, %1, %1\n\ts_add_u32 s20, s20, 1" : "+v"(a) : "v"(b) : "s20", "scc");
#define MAX3_SALU2(a, b) asm volatile("v_max3_i32 %0, %0, %1, %1\n\ts_add_u32 s20, s20, 1\n\ts_and_b64 s[22:23], s[22:23], exec" : "+v"(a) : "v"(b) : "s20", "s22", "s23", "scc");
#define ADD_NOP(a, b) asm volatile("v_add_u32 %0, %0, %1\n\ts_nop 0" : "+v"(a) : "v"(b));
#define ADD_WAIT(a, b) asm volatile("v_add_u32 %0, %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(a) : "v"(b));
#define SALU_ONLY(a, b) asm volatile("s_add_u32 s20, s20, 1" : : : "s20", "scc");
#define ADD_E64(a, b) asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(a) : "v"(b));

#define KERNEL(NAME, OP)                                                                         \
  __global__ void __launch_bounds__(256) NAME(int* out, long long* cyc, int seed) {              \
    int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    int b0 = seed, b1 = seed + 1, b2 = seed + 2, b3 = seed + 3;                                  \
    long long a0w = a0, a1w = a1, a2w = a2, a3w = a3, a4w = a4, a5w = a5, a6w = a6, a7w = a7, b0w = b0, b1w = b1, b2w = b2, b3w = b3; \
    const long long t0 = __builtin_amdgcn_s_memtime();                                           \
    for (int it = 0; it < ITER; ++it) {                                                          \
      OPS8(OP) OPS8(OP) OPS8(OP) OPS8(OP) OPS8(OP) OPS8(OP) OPS8(OP) OPS8(OP)                    \
    }                                                                                            \
    const long long t1 = __builtin_amdgcn_s_memtime();                                           \
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (int)(a0w + a1w + a2w + a3w + a4w + a5w + a6w + a7w); \
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;             \
  }

// the same stream with only some lanes enabled: does a SIMD skip the 16-lane passes that have no active lane?
#define KERNEL_EXEC(NAME, OP, MASK)                                                              \
  __global__ void __launch_bounds__(256) NAME(int* out, long long* cyc, int seed) {              \
    int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    int b0 = seed, b1 = seed + 1, b2 = seed + 2, b3 = seed + 3;                                  \
    const long long t0 = __builtin_amdgcn_s_memtime();                                           \
    asm volatile("s_mov_b64 exec, %0" ::"s"((unsigned long long)(MASK)));                        \
    for (int it = 0; it < ITER; ++it) {                                                          \
      OPS8(OP) OPS8(OP) OPS8(OP) OPS8(OP) OPS8(OP) OPS8(OP) OPS8(OP) OPS8(OP)                    \
    }                                                                                            \
    asm volatile("s_mov_b64 exec, -1");                                                          \
    const long long t1 = __builtin_amdgcn_s_memtime();                                           \
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                \
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;             \
  }
KERNEL_EXEC(k_add_exec5, ADD_U32, 0x1full)
KERNEL_EXEC(k_add_exec16, ADD_U32, 0xffffull)
KERNEL_EXEC(k_add_exec32, ADD_U32, 0xffffffffull)
KERNEL_EXEC(k_max3_exec16, MAX3_I32, 0xffffull)
KERNEL_EXEC(k_add_exec0, ADD_U32, 0ull)

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
KERNEL(k_sub_u32, SUB_U32)
KERNEL(k_min_u32, MIN_U32)
KERNEL(k_add3, ADD3_U32)
KERNEL(k_lshl_add, LSHL_ADD)
KERNEL(k_bfi, BFI_B32)
KERNEL(k_ashr, ASHR_I32)
KERNEL(k_and, AND_B32)
KERNEL(k_or, OR_B32)
KERNEL(k_and_or, AND_OR)
KERNEL(k_med3, MED3_I32)
KERNEL(k_max_u32, MAX_U32)
KERNEL(k_max_i16, MAX_I16)
KERNEL(k_max_f32, MAX_F32)
KERNEL(k_cnd_e64, CND_E64)
KERNEL(k_cmp_e64, CMP_E64)
KERNEL(k_cmp_vcc, CMP_VCC)
KERNEL(k_cmp_cnd, CMP_CND)
KERNEL(k_dpp_min, DPP_MIN)
KERNEL(k_dpp_mov, DPP_MOV)
KERNEL(k_mov_b64, MOV_B64)
KERNEL(k_subrev_co, SUBREV_CO)
KERNEL(k_mad_i24, MAD_I24)
KERNEL(k_add_sgpr, ADD_SGPR)
KERNEL(k_add_inl, ADD_INL)
KERNEL(k_add_lit, ADD_LIT)
KERNEL(k_max3_sgpr, MAX3_SGPR)
KERNEL(k_mix_add_max, MIX_ADD_MAX)
KERNEL(k_mix_add_max3, MIX_ADD_MAX3)
KERNEL(k_mix_sgpr, MIX_SGPR)
KERNEL(k_add_e64, ADD_E64)
KERNEL(k_add_salu, ADD_SALU)
KERNEL(k_max3_salu, MAX3_SALU)
KERNEL(k_max3_salu2, MAX3_SALU2)
KERNEL(k_add_nop, ADD_NOP)
KERNEL(k_add_wait, ADD_WAIT)
KERNEL(k_salu_only, SALU_ONLY)

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
      {"v_perm_b32", k_perm_b32}, {"v_or3_b32", k_or3_b32}, {"v_sub_u32", k_sub_u32}, {"v_min_u32", k_min_u32},
      {"v_add3_u32", k_add3}, {"v_lshl_add_u32", k_lshl_add}, {"v_bfi_b32", k_bfi}, {"v_ashrrev_i32", k_ashr},
      {"v_and_b32", k_and}, {"v_or_b32", k_or}, {"v_and_or_b32", k_and_or}, {"v_med3_i32", k_med3}, {"v_max_u32", k_max_u32},
      {"v_max_i16", k_max_i16}, {"v_max_f32", k_max_f32}, {"v_cndmask_e64 sgpr", k_cnd_e64}, {"v_cmp_lt e64", k_cmp_e64},
      {"v_cmp_lt vcc", k_cmp_vcc}, {"cmp+nop+cndmask", k_cmp_cnd}, {"v_min_i32_dpp", k_dpp_min}, {"v_mov_b32_dpp", k_dpp_mov},
      {"v_mov_b64", k_mov_b64}, {"v_subrev_co_u32", k_subrev_co}, {"v_mad_i32_i24", k_mad_i24},
      {"v_add_u32 sgpr", k_add_sgpr}, {"v_add_u32 inline", k_add_inl}, {"v_add_u32 literal", k_add_lit}, {"v_max3 sgpr", k_max3_sgpr},
      {"add,max (x2)", k_mix_add_max}, {"add,add,add,max3 (x4)", k_mix_add_max3}, {"addS,addS,add,max3(x4)", k_mix_sgpr},
      {"v_add_u32_e64", k_add_e64}, {"pair: v_add + s_add", k_add_salu}, {"pair: v_max3 + s_add", k_max3_salu},
      {"trio: v_max3+s_add+s_and", k_max3_salu2}, {"pair: v_add + s_nop", k_add_nop}, {"pair: v_add + s_waitcnt", k_add_wait},
      {"s_add_u32 alone", k_salu_only},
      {"v_add_u32 exec=5 lanes", k_add_exec5}, {"v_add_u32 exec=16 lanes", k_add_exec16}, {"v_add_u32 exec=32 lanes", k_add_exec32},
      {"v_max3_i32 exec=16 lanes", k_max3_exec16}, {"v_add_u32 exec=0", k_add_exec0}};
  const char* only = getenv("VALU_RATE_ONLY");  // substring filter
  printf("%-26s %10s %22s %26s\n", "instruction", "waves/SIMD", "cycles/instr (a wave)", "SIMD cycles/wave-instr");
  for (auto& kd : kinds) {
    if (only && !strstr(kd.name, only)) continue;
    for (int wps = 1; wps <= 4; ++wps) {
      if (wps == 4 && getenv("VALU_RATE_MAX3")) continue;
      const int blocks = cus * wps;  // 256-thread blocks: one wave per SIMD each; wps blocks per CU
      for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kd.k, dim3(blocks), dim3(256), 0, 0, out, cyc, rep);
      CK(hipDeviceSynchronize());
      std::vector<long long> h(4 * blocks);
      CK(hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
      std::sort(h.begin(), h.end());
      const double med = (double)h[h.size() / 2] / ((double)UNROLL * ITER);
      printf("%-26s %10d %22.2f %26.2f\n", kd.name, wps, med, med / wps);
    }
  }
  return 0;
}
