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
  if (b->affine) {  // ring of derived values (bialign_wide.hpp): WIDE_RING levels per pair of this launch
    std::vector<int64_t> off(count);
    int64_t total = 0;
    for (int t = 0; t < count; ++t) {
      off[t] = total;
      total += WIDE_RING * wide_ring_level_dwords(b->pairs[b->order[first + t]].n, b->S);
    }
    if (b->d_wide_ring.n < (size_t)total) HIP_TRY(b->d_wide_ring.alloc((size_t)total));
    if (b->d_wide_off.n < (size_t)count) HIP_TRY(b->d_wide_off.alloc((size_t)count));
    HIP_TRY(hipMemcpyAsync(b->d_wide_off.p, off.data(), sizeof(int64_t) * count, hipMemcpyHostToDevice, b->eng->stream));
    HIP_TRY(hipStreamSynchronize(b->eng->stream));  // (`off` goes out of scope; a launch per chunk, not per step of a sweep)
    w.wide_ring = b->d_wide_ring.p;
    w.wide_ring_off = b->d_wide_off.p;
    w.wide_score_only = b->lean ? 1 : 0;
  }
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
