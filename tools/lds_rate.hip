// lds_rate.hip -- what does a lane-to-lane exchange cost on gfx950: ds_bpermute_b32 (no LDS storage, the LDS crossbar)
// against a ds_write2_b32 + ds_read2_b32 round trip through an exchange array (what the fill kernels do today)?
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/lds_rate tools/lds_rate.hip && /tmp/lds_rate
// Per kind and waves per CU (4, 8, 12): cycles per exchanged dword as one wave sees them, and CU cycles per
// wave-dword (the LDS pipe is one per CU).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int ITER = 2048;

// 16 exchanged dwords per iteration
__global__ void __launch_bounds__(256) k_bpermute(int* out, long long* cyc, int seed) {
  int v[16];
  for (int x = 0; x < 16; ++x) v[x] = threadIdx.x * 17 + x + seed;
  const int addr = (((threadIdx.x & 63) + 61) & 63) * 4;  // lane L-3
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int x = 0; x < 16; ++x) v[x] = __builtin_amdgcn_ds_bpermute(addr, v[x]) + 1;
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  int s = 0;
  for (int x = 0; x < 16; ++x) s += v[x];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// the same 16 dwords through an exchange array [16][65] per wave: write own column, read column L-3
__global__ void __launch_bounds__(256) k_lds_roundtrip(int* out, long long* cyc, int seed) {
  __shared__ int xch[4][16 * 65];
  int v[16];
  for (int x = 0; x < 16; ++x) v[x] = threadIdx.x * 17 + x + seed;
  const int L = threadIdx.x & 63, w = threadIdx.x >> 6;
  int* mine = xch[w] + L;
  const int* theirs = xch[w] + ((L + 61) & 63);
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int x = 0; x < 16; ++x) mine[x * 65] = v[x];
#pragma unroll
    for (int x = 0; x < 16; ++x) v[x] = theirs[x * 65] + 1;
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  int s = 0;
  for (int x = 0; x < 16; ++x) s += v[x];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  int* out;
  long long* cyc;
  CK(hipMalloc(&out, sizeof(int) * 256 * cus * 4));
  CK(hipMalloc(&cyc, sizeof(long long) * 4 * cus * 4));
  struct { const char* name; void (*k)(int*, long long*, int); } kinds[] = {{"ds_bpermute_b32", k_bpermute},
                                                                           {"ds_write2+ds_read2", k_lds_roundtrip}};
  printf("%-20s %10s %26s %24s\n", "exchange", "waves/CU", "cycles/dword (a wave)", "CU cycles/wave-dword");
  for (auto& kd : kinds)
    for (int bpc = 1; bpc <= 3; ++bpc) {
      const int blocks = cus * bpc;
      for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kd.k, dim3(blocks), dim3(256), 0, 0, out, cyc, rep);
      CK(hipDeviceSynchronize());
      std::vector<long long> h(4 * blocks);
      CK(hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
      std::sort(h.begin(), h.end());
      const double med = (double)h[h.size() / 2] / (16.0 * ITER);
      printf("%-20s %10d %26.2f %24.2f\n", kd.name, 4 * bpc, med, med / (4 * bpc));
    }
  return 0;
}
