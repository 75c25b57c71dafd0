// bialign_wide.hip -- launchers of the wide-band path (any max_shift; kernels in bialign_wide.hpp and the
// WIDE forms of the traceback kernels).  One translation unit: the band width is a runtime value.
#include "bialign_host.hpp"

namespace bialign {

int launch_fill_wide(bialign_batch* b, const DeviceBatch& v, int first, int count) {
  DeviceBatch w = v;
  w.order = v.order + first;
  b->last_team = 16;  // one 1024-thread workgroup per pair
  if (b->affine)
    hipLaunchKernelGGL(fill_wide_affine_kernel<0>, dim3(count), dim3(1024), 0, b->eng->stream, w, b->S);
  else
    hipLaunchKernelGGL(fill_wide_linear_kernel<0>, dim3(count), dim3(1024), 0, b->eng->stream, w, b->S);
  HIP_TRY(hipGetLastError());
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
