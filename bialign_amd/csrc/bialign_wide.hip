// bialign_wide.hip -- launchers of the wide-band path (any max_shift; kernels in bialign_wide.hpp and the
// WIDE forms of the traceback kernels).  One translation unit: the band width is a runtime value.
#include "bialign_host.hpp"

namespace bialign {

int launch_fill_wide(bialign_batch* b, const DeviceBatch& v, int first, int count) {
  DeviceBatch w = v;
  w.order = v.order + first;
  b->packed_layers = false;
  const void* kern = b->affine ? reinterpret_cast<const void*>(fill_wide_affine_kernel<0>)
                               : reinterpret_cast<const void*>(fill_wide_linear_kernel<0>);
  // Workgroups ("parts") per pair: as many as keep the device busy and can all be resident at once (they meet at a
  // counter after every level), no more than a level has work for; one after a lost-co-residency recovery.
  int parts = 1;
  if (!b->no_xcu) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, WIDE_THREADS, 0) != hipSuccess) {
      (void)hipGetLastError();
      per_cu = 0;
    }
    int busiest = 1;  // points of the largest level, over the pairs of this launch: ~ min(n, m) rows x W x (W+1)/2
    for (int t = first; t < first + count; ++t) {
      const PairDesc& d = b->pairs[b->order[t]];
      const int W = 2 * b->S + 1;
      busiest = std::max<int64_t>(busiest, (int64_t)(std::min(d.n, d.m) + 1) * W * ((W + 1) / 2));
    }
    parts = std::max(1, std::min({per_cu * b->eng->num_cu / std::max(count, 1), (busiest + WIDE_THREADS - 1) / WIDE_THREADS,
                                  PROG_WORDS}));
    if (const char* e = getenv("BIALIGN_WIDE_PARTS")) parts = std::max(1, std::min(atoi(e), parts));  // tests
  }
  w.team = parts;
  w.spin_limit = b->xcu_spin_limit;
  b->last_team = parts * (WIDE_THREADS / 64) * (parts > 1 ? -1 : 1);
  if (b->d_prog.n < (size_t)count * PROG_WORDS) HIP_TRY(b->d_prog.alloc((size_t)count * PROG_WORDS));
  HIP_TRY(hipMemsetAsync(b->d_prog.p, 0, (size_t)count * PROG_WORDS * sizeof(int32_t), b->eng->stream));
  w.prog = b->d_prog.p;
  if (parts > 1) {
    b->used_xcu = true;
    if (int rc = xcu_serial_begin(b->eng)) return rc;
  }
  if (b->affine)
    hipLaunchKernelGGL(fill_wide_affine_kernel<0>, dim3(count * parts), dim3(WIDE_THREADS), 0, b->eng->stream, w, b->S);
  else
    hipLaunchKernelGGL(fill_wide_linear_kernel<0>, dim3(count * parts), dim3(WIDE_THREADS), 0, b->eng->stream, w, b->S);
  const hipError_t launched = hipGetLastError();
  if (parts > 1) {
    const int rc = xcu_serial_end(b->eng);
    if (launched == hipSuccess && rc) return rc;
  }
  HIP_TRY(launched);
  return BIALIGN_OK;
}

int launch_traceback_wide(const bialign_batch* b, const DeviceBatch& v, int first, int count, bool do_trace) {
  DeviceBatch w = v;
  w.order = v.order + first;
  auto go = [&](auto kern, size_t lds) -> int {
    if (lds > 64 * 1024)
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(count), dim3(64), lds, b->eng->stream, w, count);
    HIP_TRY(hipGetLastError());
    return BIALIGN_OK;
  };
  if (b->affine)
    return do_trace ? go(traceback_affine_kernel<0, true, false, true>, b->lds_trace)
                    : go(traceback_affine_kernel<0, false, false, true>, 0);
  return do_trace ? go(traceback_linear_kernel<0, true, false, true>, b->lds_trace)
                  : go(traceback_linear_kernel<0, false, false, true>, 0);
}

int launch_dump_wide(const bialign_batch* b, const DeviceBatch& v, int pid, int32_t* d_out) {
  if (b->affine)
    hipLaunchKernelGGL((dump_wide_kernel<9>), dim3(256), dim3(256), 0, b->eng->stream, v, b->S, pid, d_out);
  else
    hipLaunchKernelGGL((dump_wide_kernel<1>), dim3(256), dim3(256), 0, b->eng->stream, v, b->S, pid, d_out);
  HIP_TRY(hipGetLastError());
  return BIALIGN_OK;
}

}  // namespace bialign
