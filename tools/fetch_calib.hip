// fetch_calib.hip -- what does rocprofv3's FETCH_SIZE report for the ghost feed's access pattern?
//
// The fill kernels re-read strip-bottom rows by LDS-DMA (global_load_lds_dwordx4, per-lane source address) as
// 48-byte runs -- three 16-byte pieces, band rows a = -1, 0, 1 of one chunk of one record -- at the chunk pitch
// of the record layout (Rec<1,9>: 960 B per chunk, 6528 B per record; the run sits at bytes 912..959 of a chunk,
// i.e. inside ONE 64-byte sector).  MI355X_MICROARCH.md says FETCH_SIZE reports half the bytes of a WIDE 16 B/lane
// stream; this program issues a known number of pieces in both patterns, each byte of the buffer at most once, over
// a buffer far larger than the 256 MiB Infinity Cache, so that FETCH_SIZE can be set against known byte counts:
//
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/fetch_calib tools/fetch_calib.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -o calib -- /tmp/fetch_calib
//
// Kernels:  wide_stream  -- every lane 16 consecutive bytes, 1 KiB contiguous per wave instruction
//           ghost_runs   -- the ghost feed's runs (full records), 21 pieces per record
//           ghost_packed -- the same over packed records (Pack<1>: 4 chunks of 960 B per record)
// It prints per kernel: bytes requested (16 B x active lanes), distinct 64-B sectors x 64, distinct 128-B lines x 128.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void dma16(const void* p, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(p), "s"(lds_dst) : "memory");
}

__global__ void __launch_bounds__(64) wide_stream(const char* buf, int64_t kib_per_wave) {
  __shared__ __attribute__((aligned(16))) char lds[1024];
  const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds);
  const char* p = buf + ((int64_t)blockIdx.x * kib_per_wave << 10) + threadIdx.x * 16;
  for (int64_t k = 0; k < kib_per_wave; ++k) {
    dma16(p + (k << 10), dst);
    if ((k & 7) == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// one wave instruction = the 21 pieces of three consecutive records (lane 63 repeats lane 62's piece)
template <int NCHUNK, int RECB>
__global__ void __launch_bounds__(64) ghost_runs(const char* buf, int64_t triples_per_wave) {
  __shared__ __attribute__((aligned(16))) char lds[1024];
  const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds);
  const int lane = threadIdx.x < 63 ? threadIdx.x : 62;
  constexpr int PPR = 3 * NCHUNK;  // pieces per record
  const int sub = lane / PPR, piece = lane - sub * PPR, c = piece / 3, aa = piece - 3 * c;
  const bool on = threadIdx.x < 3 * PPR;  // lanes beyond three records' pieces stay idle
  const int64_t rec0 = (int64_t)blockIdx.x * triples_per_wave * 3;
  for (int64_t k = 0; k < triples_per_wave; ++k) {
    const char* p = buf + (rec0 + 3 * k + sub) * RECB + c * 960 + (57 + aa) * 16;
    if (on) dma16(p, dst);
    if ((k & 7) == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int main() {
  const int64_t bytes = (int64_t)12 << 30;
  char* buf = nullptr;
  CK(hipMalloc(&buf, bytes));
  CK(hipMemset(buf, 1, bytes));
  CK(hipDeviceSynchronize());
  const int waves = 4096;
  {
    const int64_t kib = bytes / 1024 / waves;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(wide_stream, dim3(waves), dim3(64), 0, 0, buf, kib);
    CK(hipDeviceSynchronize());
    const double req = (double)kib * 1024 * waves;
    printf("wide_stream : requested %.0f B  sectors64 %.0f B  lines128 %.0f B   (per launch, 2 launches)\n", req, req, req);
  }
  {
    constexpr int RECB = 6528, NCH = 7;  // Rec<1,9>: 6 chunks + the tail piece
    const int64_t triples = bytes / RECB / 3 / waves;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((ghost_runs<NCH, RECB>), dim3(waves), dim3(64), 0, 0, buf, triples);
    CK(hipDeviceSynchronize());
    const double runs = (double)triples * 3 * NCH * waves;
    // a run is bytes 912..959 of a 960-byte chunk at c*960 of a 6528-byte record: 6528 = 102*64, 960 = 15*64 -> one sector, one line
    printf("ghost_runs  : requested %.0f B  sectors64 %.0f B  lines128 %.0f B\n", runs * 48, runs * 64, runs * 128);
  }
  {
    constexpr int RECB = 3840, NCH = 4;  // Pack<1>: 4 chunks of 960 B
    const int64_t triples = bytes / RECB / 3 / waves;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((ghost_runs<NCH, RECB>), dim3(waves), dim3(64), 0, 0, buf, triples);
    CK(hipDeviceSynchronize());
    const double runs = (double)triples * 3 * NCH * waves;
    printf("ghost_packed: requested %.0f B  sectors64 %.0f B  lines128 %.0f B\n", runs * 48, runs * 64, runs * 128);
  }
  CK(hipFree(buf));
  return 0;
}
