// bialign_dump.hpp -- layers back into the reference's array layout.  Part of bialign_kernels.hpp (include that, not this).
#pragma once

namespace bialign {

// ---------------------------------------------------------------------------
// Layer dump in the reference layout (tests only).
// ---------------------------------------------------------------------------
template <int S, int NL, bool PACK = false>
__global__ void dump_layers_kernel(const DeviceBatch A, int pid, int32_t* out) {
  constexpr int W = 2 * S + 1;
  const PairDesc pd = A.pairs[pid];
  const int n = pd.n, m = pd.m;
  const int64_t cells = (int64_t)(n + 1) * (m + 1) * W * W;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < cells;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int bb = t % W, aa = (t / W) % W;
    const int j = (t / (W * W)) % (m + 1), i = t / ((int64_t)W * W * (m + 1));
    const int k = i + aa - S, l = j + bb - S;
    const bool ok = k >= 0 && k <= n && l >= 0 && l <= m;
    for (int q = 0; q < NL; ++q)
      out[q * cells + t] = !ok ? 0 : (PACK ? packed_cell<S>(A.layers, pd, i, j, aa, bb, q) : A.layers[cell_dword<S, NL>(pd, i, j, aa, bb, q)]);
  }
}

}  // namespace bialign
