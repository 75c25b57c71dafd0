// bialign_host.hpp -- host-side internals shared by the translation units of libbialign_hip.so:
// the batch / engine objects behind the C ABI and the kernel launchers.  The launchers are
// templates on max_shift; each (max_shift, kind) is instantiated in its own translation unit
// (bialign_inst.hip, compiled once per -DBIALIGN_TU_S / -DBIALIGN_TU_KIND) so that the kernels
// build in parallel; bialign_capi.hip only dispatches.
#pragma once
#include "bialign_kernels.hpp"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/bialign.h"

#define BIALIGN_MAX_SHIFT_PACKED 3  // packed layer records (Pack<S>) are instantiated for max_shift 1..3

namespace bialign {

int fail(int code, const char* fmt, ...);  // records the message bialign_last_error() returns

#define HIP_TRY(expr)                                                                 \
  do {                                                                                \
    hipError_t e_ = (expr);                                                           \
    if (e_ != hipSuccess)                                                             \
      return fail(BIALIGN_E_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                  __FILE__, __LINE__);                                                \
  } while (0)

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  void swap(DevBuf& o) {
    std::swap(p, o.p);
    std::swap(n, o.n);
  }
  hipError_t alloc(size_t count) {
    release();
    n = count;
    return hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(count, 1) * sizeof(T));
  }
  hipError_t upload(const T* src, size_t count, hipStream_t s) {
    hipError_t e = alloc(count);
    if (e != hipSuccess || count == 0) return e;
    return hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, s);
  }
};

// GhostFeed<S,.>::BLK
inline int ghost_blk(int S) {
#ifdef BIALIGN_BLK_OVERRIDE
  (void)S;
  return BIALIGN_BLK_OVERRIDE;
#else
  return S <= 1 ? 8 : 4;
#endif
}

}  // namespace bialign

using bialign::DevBuf;
using bialign::DeviceBatch;
using bialign::PairDesc;
using bialign::TraceState;

struct bialign_engine {
  int device = 0;
  int num_cu = 256;
  hipStream_t stream = nullptr;
  hipStream_t copy_stream = nullptr;  // uploads of new batches: not ordered behind running sweeps
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  // Layer buffer kept between batches: hipMalloc / hipFree of tens of GB cost 0.1-6 s, the
  // sweep itself ~20 ms.  A batch takes it at creation when it is large enough and hands
  // the larger of (its own, the cached one) back at destruction; bialign_engine_trim frees it.
  int live_batches = 0;   // batches created on this engine and not yet destroyed
  bool closing = false;   // bialign_engine_destroy was called while batches were alive: the last one finishes the job
  DevBuf<int32_t> layer_cache;
  DevBuf<int32_t> layer_cache2;  // second slot: filled only when two batches were alive at once (pipelined use)
};

struct bialign_batch {
  bialign_engine* eng = nullptr;
  bialign_params prm{};
  int affine = 0, NL = 1, S = 0;
  int npairs = 0;
  std::vector<PairDesc> pairs;      // host mirror (layer_off valid for the pair's chunk)
  std::vector<int32_t> order;       // chunk-by-chunk launch order
  std::vector<int> chunk_begin;     // index into order, size nchunks+1
  int64_t cells = 0, trace_bytes = 0, max_chunk_dwords = 0;
  size_t lds_bytes = 0;                   // dynamic LDS of a one-wave workgroup
  size_t lds_base = 0, lds_per_wave = 0;  // team launches: lds_base + T * lds_per_wave
  size_t lds_diet8 = 0;                   // eight-wave workgroups of the s=2 affine kernel (DIET layout)
  // fill_affine_slim_kernel (bialign_fill_slim.hpp): twelve ghost rings + tables, and per pair of the workgroup its codes
  size_t lds_slim_base = 0, lds_slim_codes = 0;
  size_t lds_slim(int tw) const { return lds_slim_base + (size_t)(12 / tw) * lds_slim_codes; }
  size_t lds_trace = 0;                   // tracebacks: score tables + sequence codes
  DevBuf<PairDesc> d_pairs;
  DevBuf<int32_t> d_order, d_s1, d_s2, d_layers, d_scores, d_tlen, d_complete, d_err, d_prog;
  int last_team = 1;  // waves per pair of the last fill launch (negative: cross-CU team)
  // Cross-CU teams need every workgroup of the launch resident at once.  The grid is sized from the
  // occupancy the runtime reports for the actual kernel (xcu_resident, cached per LEAN flavour) and
  // launches of this kind are serialised across the engines of a process; if a hand-off still times
  // out (another tenant on the device), the run is repeated with in-workgroup teams (no_xcu).
  int xcu_resident[4] = {-1, -1, -1, -1};  // [LEAN + 2 * (eight-wave workgroups)]
  bool used_xcu = false;   // a fill launch of the pending / last run was a cross-CU team
  bool no_xcu = false;     // a cross-CU launch of this batch failed once: in-workgroup teams from now on
  int recovered = 0;       // runs repeated after a hand-off timeout
  int xcu_spin_limit = 1 << 20;  // polls before a cross-CU wave gives up (~1 s); BIALIGN_XCU_SPIN_LIMIT: tests
  uint32_t pending_flags = 0;
  // Packed records (Pack<S>, bialign_types.hpp): decided per batch at creation (affine, max_shift 1 or 2, LOOKUP,
  // full storage, beta <= 0, every pair long enough that most steps are interior); dropped for good when a sweep
  // meets an offset that does not fit 16 bits (device flag bit 2 -> the run is repeated with full records).
  bool pack = false, pack_failed = false;
  bool used_pack = false;       // a fill launch of the pending / last run stored packed records
  bool packed_layers = false;   // ... and so did the launch whose layers are in the buffer now
  bool pack_now() const { return pack && !pack_failed; }
  bool packed_sizing = false;          // chunks and pair offsets were planned with the packed sizes
  std::vector<int64_t> full_dwords;    // per pair: dwords of its full-record form (for the fallback's re-plan)
  DevBuf<uint8_t> d_seq_a, d_cls_a, d_seq_b, d_cls_b, d_trace;
  DevBuf<int32_t> d_mu2;  // dense-mu2 mode: all pairs' n x m tables
  DevBuf<int32_t> d_wide_ring;  // wide-band affine sweep: derived values of the last levels (bialign_wide.hpp)
  DevBuf<int64_t> d_wide_off;   // ... per pair of a launch: offset of its ring
  bool dense = false;
  bool wide = false;        // max_shift above the tiled kernels: anti-diagonal path (bialign_wide.hpp), reference-order layers
  bool lean = false;        // LEAN records: the sweep keeps only the strip-bottom rows
  bool lean_trace = false;  // ... and tracebacks re-sweep one strip at a time into a scratch area
  DevBuf<TraceState> d_tstate;
  int resw_k = 1;           // strips re-swept and walked per round (more when the batch has few pairs)
  int k1 = 0, k2 = 0;
  bialign_timing timing{};
  bool ran = false, ran_trace = false;
  bool pending = false, pending_trace = false;  // an enqueued run not yet waited for
  std::vector<hipEvent_t> evs;                  // three per chunk: before fill, after fill, after traceback
  hipEvent_t uploaded = nullptr;                // inputs are in HBM (recorded on the copy stream)
  ~bialign_batch() {
    for (hipEvent_t e : evs)
      if (e) (void)hipEventDestroy(e);
    if (uploaded) (void)hipEventDestroy(uploaded);
  }

  DeviceBatch view() const {
    DeviceBatch v{};
    v.pairs = d_pairs.p;
    v.order = d_order.p;
    v.seq_a = d_seq_a.p; v.cls_a = d_cls_a.p; v.seq_b = d_seq_b.p; v.cls_b = d_cls_b.p;
    v.s1 = d_s1.p; v.s2 = d_s2.p;
    v.k1 = k1; v.k2 = k2;
    v.beta = prm.gap_opening_cost; v.gamma = prm.gap_cost; v.delta = prm.shift_cost;
    v.layers = d_layers.p;
    v.scores = d_scores.p;
    v.trace = d_trace.p;
    v.trace_len = d_tlen.p;
    v.complete = d_complete.p;
    v.errflag = d_err.p;
    v.mu2_dense = dense ? d_mu2.p : nullptr;
    v.scratch = d_layers.p;  // a pair's scratch records follow its LEAN records in the same buffer
    v.tstate = d_tstate.p;
    v.resw_k = resw_k;
    v.wide_s = S;
    v.prio_mode = getenv("BIALIGN_PRIO") ? atoi(getenv("BIALIGN_PRIO")) : 1;
    v.spin_limit = 1 << 20;  // waves of one workgroup are co-resident by construction: a timeout there is a bug
    return v;
  }
};

namespace bialign {

// One launch shape: TW waves per workgroup, GW workgroups per pair (GW > 1 = cross-CU team).
struct TeamShape {
  int tw = 1, gw = 1;
  bool slim = false;  // the three-waves-per-SIMD kernel (fill_affine_slim_kernel), teams of tw = 2, 3, 6 or 12 waves
  int waves() const { return tw * gw; }
};
// fill_affine_slim_kernel exists for this batch: affine, max_shift 1, LOOKUP scores, beta <= 0, packed records, full storage
inline bool slim_available(const bialign_batch* b) {
  const char* sw = getenv("BIALIGN_SLIM");  // "0": tests / A-B, the two-wave kernels only
  const bool off = sw && atoi(sw) == 0;
  return !off && b->affine && b->S == 1 && !b->dense && !b->wide && b->prm.gap_opening_cost <= 0 && (b->lean || b->pack_now());
}

// xcu_resident: one-wave workgroups of the cross-CU kernel the device holds at once (0: no such kernel);
// xcu8_resident: likewise its eight-wave workgroups (s=2 affine sweep only, else 0)
TeamShape team_shape(const bialign_batch* b, int first, int count, int xcu_resident, int xcu8_resident = 0);
inline bool diet8_available(const bialign_batch* b) {  // the eight-wave s=2 affine kernel and its LDS layout
  return b->affine && b->S == 2 && !b->dense && b->lds_diet8 <= 160 * 1024;
}
// Cross-CU launches of all engines of this process on one device run one after the other (each needs
// the whole device's wave slots): the stream waits for the previous such launch, the new one is recorded.
int xcu_serial_begin(bialign_engine* e);
int xcu_serial_end(bialign_engine* e);

template <int S, bool BETA_NONPOS, int TW, bool XCU, bool DENSE = false, bool LEAN = false, bool PACK = false>
int launch_fill_affine_t(bialign_batch* b, const DeviceBatch& v, int first, int count, int gw) {
  DeviceBatch w = v;
  w.order = v.order + first;
  w.team = gw;
  b->packed_layers = PACK;
  if (PACK) b->used_pack = true;
  auto kern = fill_affine_kernel<S, BETA_NONPOS, TW, XCU, DENSE, LEAN, false, PACK>;
  const size_t lds = (S == 2 && TW == 8) ? b->lds_diet8 : b->lds_base + (size_t)TW * b->lds_per_wave;
  if (lds > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  if (XCU) {
    if (b->d_prog.n < (size_t)count * PROG_WORDS) HIP_TRY(b->d_prog.alloc((size_t)count * PROG_WORDS));
    HIP_TRY(hipMemsetAsync(b->d_prog.p, 0, (size_t)count * PROG_WORDS * sizeof(int32_t), b->eng->stream));
    w.prog = b->d_prog.p;
    w.spin_limit = b->xcu_spin_limit;
    b->used_xcu = true;
    if (int rc = xcu_serial_begin(b->eng)) return rc;
  }
  hipLaunchKernelGGL(kern, dim3(count * (XCU ? gw : 1)), dim3(64 * TW), lds, b->eng->stream, w);
  const hipError_t launched = hipGetLastError();
  if (XCU) {
    const int rc = xcu_serial_end(b->eng);  // always: it releases the launch lock
    if (launched == hipSuccess && rc) return rc;
  }
  HIP_TRY(launched);
  return BIALIGN_OK;
}

template <int S, int TW, bool LEAN>
int launch_fill_affine_slim_t(bialign_batch* b, const DeviceBatch& v, int first, int count) {
  constexpr int PPW = 12 / TW;  // pairs per twelve-wave workgroup
  DeviceBatch w = v;
  w.order = v.order + first;
  w.team = 1;
  w.launch_pairs = count;
  w.slim_code_bytes = (int32_t)b->lds_slim_codes;
  b->packed_layers = !LEAN;
  if (!LEAN) b->used_pack = true;
  auto kern = fill_affine_slim_kernel<S, TW, PPW, LEAN>;
  const size_t lds = b->lds_slim(TW);
  if (lds > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((count + PPW - 1) / PPW), dim3(64 * 12), lds, b->eng->stream, w);
  HIP_TRY(hipGetLastError());
  return BIALIGN_OK;
}

template <int S, bool LEAN>
int launch_fill_affine_slim(bialign_batch* b, const DeviceBatch& v, int first, int count, int tw) {
  if constexpr (S == 1) {
    switch (tw) {
      case 2: return launch_fill_affine_slim_t<S, 2, LEAN>(b, v, first, count);
      case 3: return launch_fill_affine_slim_t<S, 3, LEAN>(b, v, first, count);
      case 6: return launch_fill_affine_slim_t<S, 6, LEAN>(b, v, first, count);
      case 12: return launch_fill_affine_slim_t<S, 12, LEAN>(b, v, first, count);
    }
  }
  return fail(BIALIGN_E_UNSUPPORTED, "no three-waves-per-SIMD sweep for max_shift %d, team %d", S, tw);
}

// One-wave workgroups of the cross-CU kernel <S, LEAN> the device can hold at once, from the runtime's
// occupancy calculation for the actual code object (registers, LDS): the cap of a cross-CU grid.
template <int S, bool LEAN, int TW = 1, bool DENSE = false>
int xcu_resident_blocks(bialign_batch* b) {
  int& cached = b->xcu_resident[(LEAN ? 1 : 0) + (TW == 8 ? 2 : 0)];  // (a batch is either DENSE or not)
  if (cached >= 0) return cached;
  cached = 0;
  if constexpr ((DENSE ? S <= 3 : true) && (TW == 1 || S == 2)) {
    if (TW == 8 && !diet8_available(b)) return cached;
    auto kern = fill_affine_kernel<S, true, TW, true, DENSE, LEAN>;
    const size_t lds = TW == 8 ? b->lds_diet8 : b->lds_base + b->lds_per_wave;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      (void)hipGetLastError();
      return cached;
    }
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 64 * TW, lds) != hipSuccess) {
      (void)hipGetLastError();
      return cached;
    }
    cached = per_cu * b->eng->num_cu;
  }
  return cached;
}

template <int S, bool LEAN>
int launch_fill_affine_l(bialign_batch* b, const DeviceBatch& v, int first, int count) {
  if (b->prm.gap_opening_cost > 0) {  // rare: general-beta algebra, one wave per pair
    b->last_team = 1;
    return b->dense ? launch_fill_affine_t<S, false, 1, false, true, LEAN>(b, v, first, count, 1)
                    : launch_fill_affine_t<S, false, 1, false, false, LEAN>(b, v, first, count, 1);
  }
  const bool xcu_ok = !b->no_xcu;
  const TeamShape ts = b->dense ? team_shape(b, first, count, xcu_ok ? xcu_resident_blocks<S, LEAN, 1, true>(b) : 0, 0)
                                : team_shape(b, first, count, xcu_ok ? xcu_resident_blocks<S, LEAN>(b) : 0,
                                             xcu_ok ? xcu_resident_blocks<S, LEAN, 8>(b) : 0);
  b->last_team = ts.waves() * (ts.gw > 1 ? -1 : 1);
  if (ts.slim) return launch_fill_affine_slim<S, LEAN>(b, v, first, count, ts.tw);
  if (b->dense) {  // dense-mu2 kernels: one-wave cross-CU teams, in-workgroup teams of 4 and 2, one wave
    if constexpr (S >= 1 && S <= BIALIGN_MAX_SHIFT_PACKED && !LEAN) {
      if (b->pack_now()) {
        if (ts.gw > 1) return launch_fill_affine_t<S, true, 1, true, true, false, true>(b, v, first, count, ts.gw);
        b->last_team = std::min(ts.tw, 4);
        if (ts.tw >= 4) return launch_fill_affine_t<S, true, 4, false, true, false, true>(b, v, first, count, 1);
        if (ts.tw >= 2) return launch_fill_affine_t<S, true, 2, false, true, false, true>(b, v, first, count, 1);
        return launch_fill_affine_t<S, true, 1, false, true, false, true>(b, v, first, count, 1);
      }
    }
    if constexpr (S <= 3) {
      if (ts.gw > 1) return launch_fill_affine_t<S, true, 1, true, true, LEAN>(b, v, first, count, ts.gw);
      b->last_team = std::min(ts.tw, 4);
      if (ts.tw >= 4) return launch_fill_affine_t<S, true, 4, false, true, LEAN>(b, v, first, count, 1);
      if (ts.tw >= 2) return launch_fill_affine_t<S, true, 2, false, true, LEAN>(b, v, first, count, 1);
    }
    b->last_team = 1;
    return launch_fill_affine_t<S, true, 1, false, true, LEAN>(b, v, first, count, 1);
  }
  if constexpr (S >= 1 && S <= BIALIGN_MAX_SHIFT_PACKED && !LEAN) {
    if (b->pack_now()) {  // same launch shapes, packed records
      if constexpr (S == 2) {
        if (ts.gw > 1 && ts.tw == 8) return launch_fill_affine_t<S, true, 8, true, false, false, true>(b, v, first, count, ts.gw);
      }
      if (ts.gw > 1) return launch_fill_affine_t<S, true, 1, true, false, false, true>(b, v, first, count, ts.gw);
      if constexpr (S <= 2) {
        if (ts.tw == 8) return launch_fill_affine_t<S, true, 8, false, false, false, true>(b, v, first, count, 1);
      }
      if (ts.tw >= 4) return launch_fill_affine_t<S, true, 4, false, false, false, true>(b, v, first, count, 1);
      if (ts.tw >= 2) return launch_fill_affine_t<S, true, 2, false, false, false, true>(b, v, first, count, 1);
      return launch_fill_affine_t<S, true, 1, false, false, false, true>(b, v, first, count, 1);
    }
  }
  if constexpr (S == 2) {
    if (ts.gw > 1 && ts.tw == 8) return launch_fill_affine_t<S, true, 8, true, false, LEAN>(b, v, first, count, ts.gw);
  }
  if (ts.gw > 1) return launch_fill_affine_t<S, true, 1, true, false, LEAN>(b, v, first, count, ts.gw);
  if constexpr (S <= 2) {  // (s=2: the DIET layout)
    if (ts.tw == 8) return launch_fill_affine_t<S, true, 8, false, false, LEAN>(b, v, first, count, 1);
  }
  if constexpr (S <= 3) {  // s >= 4 needs nearly all 512 registers of a SIMD lane: one wave per pair
    if (ts.tw >= 4) return launch_fill_affine_t<S, true, 4, false, false, LEAN>(b, v, first, count, 1);
    if (ts.tw >= 2) return launch_fill_affine_t<S, true, 2, false, false, LEAN>(b, v, first, count, 1);
  }
  return launch_fill_affine_t<S, true, 1, false, false, LEAN>(b, v, first, count, 1);
}

template <int S>
int launch_fill_affine(bialign_batch* b, const DeviceBatch& v, int first, int count) {
  return b->lean ? launch_fill_affine_l<S, true>(b, v, first, count) : launch_fill_affine_l<S, false>(b, v, first, count);
}

// Lean traceback, one round: re-sweep the strip every unfinished pair's walk stands in ...
template <int S>
int launch_resweep_affine(bialign_batch* b, const DeviceBatch& v, int first, int count) {
  DeviceBatch w = v;
  w.order = v.order + first;
  w.team = 1;
  const size_t lds = b->lds_base + b->lds_per_wave;
  auto go = [&](auto kern) -> int {
    if (lds > 64 * 1024)
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(count * b->resw_k), dim3(64), lds, b->eng->stream, w);
    HIP_TRY(hipGetLastError());
    return BIALIGN_OK;
  };
  if (b->dense)
    return b->prm.gap_opening_cost > 0 ? go(fill_affine_kernel<S, false, 1, false, true, false, true>)
                                       : go(fill_affine_kernel<S, true, 1, false, true, false, true>);
  return b->prm.gap_opening_cost > 0 ? go(fill_affine_kernel<S, false, 1, false, false, false, true>)
                                     : go(fill_affine_kernel<S, true, 1, false, false, false, true>);
}

// ... then walk through it.
template <int S>
int launch_traceback_affine_strip(const bialign_batch* b, const DeviceBatch& v, int first, int count) {
  DeviceBatch w = v;
  w.order = v.order + first;
  auto kern = traceback_affine_kernel<S, true, true>;
  if (b->lds_trace > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_trace));
  hipLaunchKernelGGL(kern, dim3(count), dim3(64), b->lds_trace, b->eng->stream, w, count);
  HIP_TRY(hipGetLastError());
  return BIALIGN_OK;
}

template <int S>
int launch_traceback_affine(const bialign_batch* b, const DeviceBatch& v, int first, int count,
                            bool do_trace) {
  DeviceBatch w = v;
  w.order = v.order + first;
  const int blocks = count;  // one wave per pair
  if (b->lds_trace > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(traceback_affine_kernel<S, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_trace));
  if constexpr (S >= 1 && S <= BIALIGN_MAX_SHIFT_PACKED) {
    if (b->packed_layers) {
      if (b->lds_trace > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(traceback_affine_kernel<S, true, false, false, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_trace));
      if (do_trace)
        hipLaunchKernelGGL((traceback_affine_kernel<S, true, false, false, true>), dim3(blocks), dim3(64), b->lds_trace,
                           b->eng->stream, w, count);
      else
        hipLaunchKernelGGL((traceback_affine_kernel<S, false, false, false, true>), dim3(blocks), dim3(64), 0,
                           b->eng->stream, w, count);
      HIP_TRY(hipGetLastError());
      return BIALIGN_OK;
    }
  }
  if (do_trace)
    hipLaunchKernelGGL((traceback_affine_kernel<S, true>), dim3(blocks), dim3(64), b->lds_trace, b->eng->stream, w, count);
  else
    hipLaunchKernelGGL((traceback_affine_kernel<S, false>), dim3(blocks), dim3(64), 0, b->eng->stream, w, count);
  HIP_TRY(hipGetLastError());
  return BIALIGN_OK;
}

template <int S, int TW, bool DENSE = false, bool LEAN = false, bool XCU = false>
int launch_fill_linear_t(bialign_batch* b, const DeviceBatch& v, int first, int count, int gw = 1) {
  DeviceBatch w = v;
  w.order = v.order + first;
  w.team = gw;
  b->packed_layers = false;
  auto kern = fill_linear_kernel<S, TW, DENSE, LEAN, false, XCU>;
  const size_t lds = b->lds_base + (size_t)TW * b->lds_per_wave;
  if (lds > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  if (XCU) {  // as in launch_fill_affine_t
    if (b->d_prog.n < (size_t)count * PROG_WORDS) HIP_TRY(b->d_prog.alloc((size_t)count * PROG_WORDS));
    HIP_TRY(hipMemsetAsync(b->d_prog.p, 0, (size_t)count * PROG_WORDS * sizeof(int32_t), b->eng->stream));
    w.prog = b->d_prog.p;
    w.spin_limit = b->xcu_spin_limit;
    b->used_xcu = true;
    if (int rc = xcu_serial_begin(b->eng)) return rc;
  }
  hipLaunchKernelGGL(kern, dim3(count * (XCU ? gw : 1)), dim3(64 * TW), lds, b->eng->stream, w);
  const hipError_t launched = hipGetLastError();
  if (XCU) {
    const int rc = xcu_serial_end(b->eng);
    if (launched == hipSuccess && rc) return rc;
  }
  HIP_TRY(launched);
  return BIALIGN_OK;
}

// one-wave workgroups of the one-layer cross-CU kernel the device holds at once
template <int S, bool LEAN, bool DENSE = false>
int xcu_resident_linear(bialign_batch* b) {
  int& cached = b->xcu_resident[LEAN ? 1 : 0];
  if (cached >= 0) return cached;
  cached = 0;
  auto kern = fill_linear_kernel<S, 1, DENSE, LEAN, false, true>;
  const size_t lds = b->lds_base + b->lds_per_wave;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
    (void)hipGetLastError();
    return cached;
  }
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 64, lds) != hipSuccess) {
    (void)hipGetLastError();
    return cached;
  }
  return cached = per_cu * b->eng->num_cu;
}

template <int S, bool LEAN>
int launch_fill_linear_l(bialign_batch* b, const DeviceBatch& v, int first, int count) {
  const bool xcu_ok = !b->no_xcu;
  const TeamShape ts = team_shape(b, first, count, !xcu_ok ? 0 : (b->dense ? xcu_resident_linear<S, LEAN, true>(b)
                                                                             : xcu_resident_linear<S, LEAN>(b)));
  b->last_team = ts.waves() * (ts.gw > 1 ? -1 : 1);
  if (b->dense) {
    if (ts.gw > 1) return launch_fill_linear_t<S, 1, true, LEAN, true>(b, v, first, count, ts.gw);
    return ts.tw >= 2 ? launch_fill_linear_t<S, 2, true, LEAN>(b, v, first, count)
                      : launch_fill_linear_t<S, 1, true, LEAN>(b, v, first, count);
  }
  if (ts.gw > 1) return launch_fill_linear_t<S, 1, false, LEAN, true>(b, v, first, count, ts.gw);
  switch (ts.tw) {
    case 8: return launch_fill_linear_t<S, 8, false, LEAN>(b, v, first, count);
    case 4: return launch_fill_linear_t<S, 4, false, LEAN>(b, v, first, count);
    case 2: return launch_fill_linear_t<S, 2, false, LEAN>(b, v, first, count);
    default: return launch_fill_linear_t<S, 1, false, LEAN>(b, v, first, count);
  }
}

template <int S>
int launch_fill_linear(bialign_batch* b, const DeviceBatch& v, int first, int count) {
  return b->lean ? launch_fill_linear_l<S, true>(b, v, first, count) : launch_fill_linear_l<S, false>(b, v, first, count);
}

template <int S>
int launch_resweep_linear(bialign_batch* b, const DeviceBatch& v, int first, int count) {
  DeviceBatch w = v;
  w.order = v.order + first;
  const size_t lds = b->lds_base + b->lds_per_wave;
  auto go = [&](auto kern) -> int {
    if (lds > 64 * 1024)
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(count * b->resw_k), dim3(64), lds, b->eng->stream, w);
    HIP_TRY(hipGetLastError());
    return BIALIGN_OK;
  };
  return b->dense ? go(fill_linear_kernel<S, 1, true, false, true>) : go(fill_linear_kernel<S, 1, false, false, true>);
}

template <int S>
int launch_traceback_linear_strip(const bialign_batch* b, const DeviceBatch& v, int first, int count) {
  DeviceBatch w = v;
  w.order = v.order + first;
  auto kern = traceback_linear_kernel<S, true, true>;
  if (b->lds_trace > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_trace));
  hipLaunchKernelGGL(kern, dim3(count), dim3(64), b->lds_trace, b->eng->stream, w, count);
  HIP_TRY(hipGetLastError());
  return BIALIGN_OK;
}

template <int S>
int launch_traceback_linear(const bialign_batch* b, const DeviceBatch& v, int first, int count,
                            bool do_trace) {
  DeviceBatch w = v;
  w.order = v.order + first;
  const int blocks = count;  // one wave per pair
  if (b->lds_trace > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(traceback_linear_kernel<S, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_trace));
  if (do_trace)
    hipLaunchKernelGGL((traceback_linear_kernel<S, true>), dim3(blocks), dim3(64), b->lds_trace, b->eng->stream, w, count);
  else
    hipLaunchKernelGGL((traceback_linear_kernel<S, false>), dim3(blocks), dim3(64), 0, b->eng->stream, w, count);
  HIP_TRY(hipGetLastError());
  return BIALIGN_OK;
}

template <int S, int NL>
int launch_dump(const bialign_batch* b, const DeviceBatch& v, int pid, int32_t* d_out) {
  if constexpr (S >= 1 && S <= BIALIGN_MAX_SHIFT_PACKED && NL == 9) {
    if (b->packed_layers) {
      hipLaunchKernelGGL((dump_layers_kernel<S, NL, true>), dim3(256), dim3(256), 0, b->eng->stream, v, pid, d_out);
      HIP_TRY(hipGetLastError());
      return BIALIGN_OK;
    }
  }
  hipLaunchKernelGGL((dump_layers_kernel<S, NL>), dim3(256), dim3(256), 0, b->eng->stream, v, pid, d_out);
  HIP_TRY(hipGetLastError());
  return BIALIGN_OK;
}

// ---- wide-band path (max_shift above BIALIGN_MAX_SHIFT_TILED, bialign_wide.hpp): runtime band width,
//      one translation unit (bialign_wide.hip) for all of it
int launch_fill_wide(bialign_batch* b, const DeviceBatch& v, int first, int count);
int launch_traceback_wide(const bialign_batch* b, const DeviceBatch& v, int first, int count, bool do_trace);
int launch_dump_wide(const bialign_batch* b, const DeviceBatch& v, int pid, int32_t* d_out);

// ---- instantiation plan: kind 0 = affine fill (the big kernels), kind 1 = everything else
#define BIALIGN_INST_KIND0(S, X)                                                                \
  X template int launch_fill_affine<S>(bialign_batch*, const DeviceBatch&, int, int);          \
  X template int launch_resweep_affine<S>(bialign_batch*, const DeviceBatch&, int, int);
#define BIALIGN_INST_KIND1(S, X)                                                                         \
  X template int launch_fill_linear<S>(bialign_batch*, const DeviceBatch&, int, int);                   \
  X template int launch_traceback_affine<S>(const bialign_batch*, const DeviceBatch&, int, int, bool);  \
  X template int launch_traceback_affine_strip<S>(const bialign_batch*, const DeviceBatch&, int, int);  \
  X template int launch_traceback_linear<S>(const bialign_batch*, const DeviceBatch&, int, int, bool);  \
  X template int launch_resweep_linear<S>(bialign_batch*, const DeviceBatch&, int, int);                \
  X template int launch_traceback_linear_strip<S>(const bialign_batch*, const DeviceBatch&, int, int);  \
  X template int launch_dump<S, 9>(const bialign_batch*, const DeviceBatch&, int, int32_t*);            \
  X template int launch_dump<S, 1>(const bialign_batch*, const DeviceBatch&, int, int32_t*);
#define BIALIGN_FOR_EACH_S(M, X) M(0, X) M(1, X) M(2, X) M(3, X) M(4, X) M(5, X)

#ifndef BIALIGN_TU_S  // every other unit: the instantiations live elsewhere
BIALIGN_FOR_EACH_S(BIALIGN_INST_KIND0, extern)
BIALIGN_FOR_EACH_S(BIALIGN_INST_KIND1, extern)
#endif

}  // namespace bialign
