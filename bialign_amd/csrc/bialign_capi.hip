// bialign_capi.hip -- C ABI (include/bialign.h) over the gfx950 kernels.
//
// Host-side responsibilities: validate, lay the batch out in HBM, cut it into
// HBM-budgeted chunks, launch fill + traceback per chunk on the engine's
// stream, time the kernels with HIP events, hand results back.  No CPU compute
// path exists here: if the device or a kernel is unavailable the call fails.
#include "bialign_host.hpp"
#include <cmath>

using namespace bialign;

namespace bialign {

static thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

// Waves per pair.  More waves per pair = more waves per SIMD when a launch has fewer pairs than
// the chip has wave slots worth filling (256 CUs x 4 SIMDs x 2).  Wave w trails wave w-1 by
// `lag` steps and wave 0 may lead wave T-1 by at most P - lag, so T waves run without mutual
// waiting only if T*lag (+ margin) fits into P; every wave should also own at least two strips.
//  * in-workgroup teams (progress words in LDS): s<=1 kernels fit 2 waves/SIMD (TW<=8), s=2,3
//    need a whole SIMD's registers per wave (TW<=4), s>=4 one wave; LDS <= 160 KB per workgroup.
//  * cross-CU teams (one-wave workgroups, progress words in HBM, write-through stores): up to 32
//    waves per pair, used when even the largest in-workgroup team leaves most SIMDs idle (few,
//    long pairs).  Every workgroup of the launch must be resident at once (a wave spins on its
//    predecessor), so the grid is capped by the residency the runtime's occupancy calculation gives
//    for the actual kernel (xcu_resident; 0 = cross-CU teams not available for this launch).
TeamShape team_shape(const bialign_batch* b, int first, int count, int xcu_resident, int xcu8_resident) {
  TeamShape ts;
  const int W = 2 * b->S + 1, R = 64 / W;
  const int lag = 2 * (R - 1) + 2 * ghost_blk(b->S) + 16;
  int fit_exact = PROG_WORDS;  // largest team the pairs of this launch allow: T*lag + 64 <= P (P >= 256), two strips per wave
  for (int t = first; t < first + count; ++t) {
    const PairDesc& d = b->pairs[b->order[t]];
    const int by_period = d.P >= 256 ? (d.P - 64) / lag : 1;
    fit_exact = std::max(1, std::min(fit_exact, std::min(by_period, d.NS / 2)));
  }
  int fit = 1;  // in-workgroup teams come in powers of two (kernel template parameter)
  while (fit * 2 <= fit_exact) fit *= 2;
  // LDS of a workgroup of t waves (the eight-wave s=2 affine kernel has its own, leaner layout)
  const bool diet8 = diet8_available(b);
  auto lds_of = [&](int t) { return (t == 8 && diet8) ? b->lds_diet8 : b->lds_base + (size_t)t * b->lds_per_wave; };
  // the one-layer (non-affine) kernel is small in registers at every s; the affine one fits two waves per
  // SIMD up to s=2 (s=2: eight waves only in the diet layout), one at s=3, and needs the whole SIMD beyond
  int tw = std::min(fit, !b->affine ? 8 : (b->S <= 1 ? 8 : (b->S == 2 ? (diet8 ? 8 : 4) : (b->S == 3 ? 4 : 1))));
  while (tw > 1 && lds_of(tw) > 160 * 1024) tw >>= 1;
  if (b->dense) tw = std::min(tw, b->affine ? 4 : 2);  // dense-mu2 kernels: up to 4 waves (affine), 2 (one layer) per workgroup
  // cross-CU teams (affine LOOKUP kernels only) take any size: the team is a runtime value there
  int gw = ((!b->affine || b->S <= 3 || !b->dense) && xcu_resident > 0) ? fit_exact : 1;  // (dense affine kernels: s <= 3)
  gw = std::max(1, std::min(gw, xcu_resident / std::max(count, 1)));
  // ... and, for the s=2 sweep, teams of eight-wave workgroups (one per CU, two waves per SIMD)
  int gw8 = (diet8 && xcu8_resident > 0) ? std::min(fit_exact / 8, xcu8_resident / std::max(count, 1)) : 0;

  const char* e = getenv("BIALIGN_TEAM");  // experiments / tests: "N" in-workgroup, "xN" cross-CU, "hN" N eight-wave workgroups
  if (e && !*e) e = nullptr;
  if (e && e[0] == 'x') {
    ts.gw = std::max(1, std::min(atoi(e + 1), gw));
    return ts;
  }
  if (e && e[0] == 'h') {
    if (gw8 >= 1 && tw == 8) {
      ts.tw = 8;
      ts.gw = std::max(1, std::min(atoi(e + 1), gw8));
    }
    return ts;
  }
  // in-workgroup: the smallest team that (nearly) maximises the waves running at once, given
  // how many workgroups of that size a CU holds (LDS, registers)
  // (registers: the one-layer kernels and the affine s=0 kernel (56) fit four waves per SIMD -- and four are measurably
  //  better than three for them, tools/occupancy_probe.py --, affine s=1 188-200 = two, counted as three here since round 1)
  const int waves_cu_regs = !b->affine ? 16 : (b->S == 0 ? 16 : (b->S == 1 ? 12 : (b->S == 2 ? 8 : 4)));
  auto concurrent = [&](int t) {
    const size_t lds = (lds_of(t) + 1023) / 1024 * 1024;
    const int wg_cu = (int)std::min<size_t>((160 * 1024) / lds, (size_t)(waves_cu_regs / t));
    // one workgroup per CU and more workgroups than CUs: they run in rounds, the last one partly empty (300 pairs x len 1024
    // as eight-wave workgroups: two rounds, 25.2 ms; cross-CU teams of six one-wave workgroups 20.1)
    if (wg_cu == 1 && count > b->eng->num_cu) return (int64_t)count * t / ((count + b->eng->num_cu - 1) / b->eng->num_cu);
    return std::min<int64_t>((int64_t)count * t, (int64_t)b->eng->num_cu * wg_cu * t);
  };
  // The three-waves-per-SIMD sweep (fill_affine_slim_kernel: 168 registers, no exchange array): teams of 2, 3, 6 or 12
  // waves in workgroups of twelve, one per CU.  Taken whenever it keeps at least as many waves running as the two-wave
  // kernels' best shape -- a SIMD runs three such waves at the per-wave speed of two (tools/valu_rate.hip).
  if (slim_available(b) && !(e && (e[0] == 'x' || e[0] == 'h'))) {
    // a workgroup = 12 waves = (12 / t) pairs x teams of t, one per CU: every SIMD holds exactly three waves
    auto conc_slim = [&](int t) { return std::min<int64_t>((int64_t)count * t, (int64_t)b->eng->num_cu * 12); };
    auto fits = [&](int t) { return t <= fit_exact && b->lds_slim(t) <= 160 * 1024; };
    auto slim_rounds = [&](int t) { return (((int64_t)count * t + 11) / 12 + b->eng->num_cu - 1) / b->eng->num_cu; };
    auto slim_score = [&](int t) {  // waves at work, averaged over the launch
      int64_t strips = 0, slots = 0;
      for (int p = first; p < first + count; ++p) {
        const int ns = b->pairs[b->order[p]].NS;
        strips += ns;
        slots += (int64_t)(ns + t - 1) / t * t;
      }
      return (double)count * t / slim_rounds(t) * strips / std::max<int64_t>(slots, 1);
    };
    static const int sizes[] = {2, 3, 6, 12};  // (a one-wave team spills in hipcc's allocation: 168 registers + scratch)
    int pick = 0;
    if (e) {  // forced in-workgroup team: the slim kernel if it comes in that size
      const int want = atoi(e);
      for (int t : sizes)
        if (t == want && fits(t)) pick = t;
    } else {
      // the team that keeps most waves at work over the launch: workgroups beyond one per CU run in rounds (all pairs of a
      // launch sweep about equally long), and a team of t idles in a pair's last round unless t divides its strips
      // (2048 pairs x len 512: teams of 2 = 342 workgroups = two rounds, the second a third full, 33.8 ms; teams of 3 =
      // two full rounds, 25.7 ms.  1280 pairs: teams of 2 in one round 16.5 ms, teams of 3 in two 21.8)
      double best_s = 0;
      for (int t : sizes)
        if (fits(t)) best_s = std::max(best_s, slim_score(t));
      for (int t : sizes)
        if (!pick && fits(t) && slim_score(t) >= best_s * 0.98) pick = t;
    }
    // what the two-wave kernels' in-workgroup teams keep running at best -- at the two waves per SIMD their registers
    // really allow (concurrent() counts three, a round-1 calibration of the choice AMONG those kernels)
    int64_t best_old = 0;
    for (int c = 1; c <= tw; c *= 2) {
      const size_t lds = (lds_of(c) + 1023) / 1024 * 1024;
      const int wg_cu = (int)std::min<size_t>((160 * 1024) / lds, (size_t)std::max(1, 8 / c));
      best_old = std::max(best_old, std::min<int64_t>((int64_t)count * c, (int64_t)b->eng->num_cu * wg_cu * c));
    }
    if (pick && !e && conc_slim(pick) < best_old) pick = 0;  // (e.g. 256 pairs whose period admits teams of 6: 1536 waves against 2048)
    // More pairs than one round of twelve-wave workgroups holds: the two-wave kernel sweeps them with one wave each, every
    // strip count divides, and workgroups of one wave refill a CU as they finish.  Three slim waves do the work of 2.06
    // two-wave ones on a SIMD (headline shape: 46.0 against 46.5 ms at strip efficiencies 0.96 and 0.98); a fractional
    // last round of one-wave workgroups costs about half a round (3072 pairs x len 512: 13.0 ms per 1024 against 11.4 at
    // 2048).  Measured, ms per 1024 pairs x len 512, slim / two-wave: 2048 pairs 12.8 / 11.4, 3072 11.8 / 13.0, 4096 12.4 / 11.3
    // (profiles/r03w_exchange/slim_rounds_512.log).
    if (pick && !e && slim_rounds(pick) > 1) {
      const double x = std::max(1.0, (double)count / (b->eng->num_cu * 8.0));  // rounds of one-wave workgroups, two per SIMD
      const double old_score = count / ((std::ceil(x) + x) / 2);
      if (slim_score(pick) * (2.06 / 3) < old_score) pick = 0;
    }
    if (pick) {
      // A handful of long pairs still go to cross-CU teams of the two-wave kernel below when that spreads them wider: a
      // third wave on a SIMD adds a few percent, an idle CU costs all of it (117 pairs x len 1024: teams of 12 on 117 CUs
      // 11.0 ms, cross-CU teams of 13 one-wave workgroups on all CUs 9.7).  Three slim waves count as 2.06 two-wave ones.
      const double run_s = conc_slim(pick) * (2.06 / 3);
      const int g = std::min(gw, std::max(1, 2048 / count));
      if (e || !(g >= 2 && (double)count * g >= run_s * 1.4)) {
        ts.tw = pick;
        ts.slim = true;
        return ts;
      }
    }
  }
  if (e) {
    int want = atoi(e), t = 1;
    while (t * 2 <= want && t * 2 <= tw) t *= 2;
    ts.tw = t;
    return ts;
  }
  // Two-wave workgroups of the s=2 affine kernel (256 registers, two such workgroups per CU) measured
  // 20-35 % slower per pair than one- or four-wave ones at the same number of resident waves
  // (tools/team_table.sh; not so at s=1 or s=3), so that sweep goes 1 -> 4.
  const bool skip2 = b->affine && b->S == 2 && tw >= 4;
  int64_t best = 0;
  for (int c = 1; c <= tw; c *= 2)
    if (!(skip2 && c == 2)) best = std::max(best, concurrent(c));
  int t = 1;
  while (t < tw && concurrent(t) * 100 < best * 95) t *= (skip2 && t == 1) ? 4 : 2;
  ts.tw = t;
  // cross-CU: when that keeps at least 1.4 x the waves running (117 pairs x len 1024: 16 one-wave workgroups per pair
  // instead of 8 waves in one, 12.9 -> 9.7 ms; 300 x len 512: 6 instead of 4, 7.6 -> 6.7 ms; at equal wave counts the
  // in-workgroup team wins: 256 x len 1024, 15.4 vs 16.8 ms) -- or, for a handful of pairs, not more waves but spread:
  // eight waves on eight CUs beat eight waves sharing one CU's SIMDs two by two (one 928 x 933 pair: 7.6 vs 9.4 ms)
  int64_t running = concurrent(t);
  {
    const int g = std::min(gw, std::max(1, 2048 / count));
    // (s=1 affine, the in-workgroup shape leaving a third of the wave slots empty: 1.2 x is enough -- 300 pairs x len 1024 as
    //  teams of 4 in one workgroup 23.3 ms, as eight-wave workgroups in two rounds 25.2, as cross-CU teams of 5 19.9)
    const bool sparse_s1 = b->affine && b->S == 1 && !b->dense && running * 100 < 2048 * 65;
    if (g >= 2 && ((int64_t)count * g * 10 >= running * (sparse_s1 ? 12 : 14) || (t == 8 && g >= 8 && count * 8 <= b->eng->num_cu))) {
      ts.tw = 1;
      ts.gw = g;
      running = (int64_t)count * g;
    }
  }
  // s=2: eight-wave workgroups spread over CUs when that keeps more waves running than either of the above
  // (64 pairs x len 2000: 4 workgroups per pair = 2048 waves, two per SIMD, against 1024 one-wave workgroups)
  if (gw8 >= 2 && (int64_t)count * gw8 * 8 * 100 >= running * 125) {
    ts.tw = 8;
    ts.gw = gw8;
  }
  return ts;
}

// ---- cross-CU launches, one at a time per device (all engines of the process)
static std::mutex g_xcu_mu;
static hipEvent_t g_xcu_done[64] = {};  // per device: the last cross-CU launch (never destroyed: process lifetime)

int xcu_serial_begin(bialign_engine* e) {
  g_xcu_mu.lock();  // held until xcu_serial_end: wait, launch and record are one step
  if (getenv("BIALIGN_XCU_NOSERIAL")) return BIALIGN_OK;  // tests: provoke lost co-residency
  hipEvent_t& ev = g_xcu_done[e->device & 63];
  hipError_t err = hipSuccess;
  if (!ev) err = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  else err = hipStreamWaitEvent(e->stream, ev, 0);
  if (err != hipSuccess) {
    g_xcu_mu.unlock();
    return fail(BIALIGN_E_DEVICE, "cross-CU launch ordering: %s", hipGetErrorString(err));
  }
  return BIALIGN_OK;
}

int xcu_serial_end(bialign_engine* e) {
  hipError_t err = hipSuccess;
  if (!getenv("BIALIGN_XCU_NOSERIAL")) err = hipEventRecord(g_xcu_done[e->device & 63], e->stream);
  g_xcu_mu.unlock();
  if (err != hipSuccess) return fail(BIALIGN_E_DEVICE, "cross-CU launch ordering: %s", hipGetErrorString(err));
  return BIALIGN_OK;
}

}  // namespace bialign

namespace {

// Sweep geometry of one pair (mirrors the kernel's Geo<S>): strips, period, steps.
void sweep_geometry(int n, int m, int S, int* NS, int* P, int* G) {
  const int W = 2 * S + 1, R = 64 / W, RR = R - 1;
  const int min_goff = 2 * ghost_blk(S) + 8;  // GhostFeed<S,.>::MIN_GOFF
  *NS = (n + 1 + RR - 1) / RR;
  // one idle column between strips (P >= m+2) and ghost records old enough to prefetch
  *P = std::max(m + 2, 2 * (R - 1) + min_goff);
  *G = (*NS - 1) * *P + m + 2 * (R - 1) + (W - 1) + 1;
}

int64_t cells_of(int n, int m, int s) {
  auto K = [s](int x) {
    int64_t t = 0;
    for (int i = 0; i <= x; ++i) t += std::min(x, i + s) - std::max(0, i - s) + 1;
    return t;
  };
  return K(n) * K(m);
}

// Pack<S> geometry for a runtime max_shift (packed records exist for max_shift 1..3)
struct PackInfo {
  int lo;
  int64_t full_recdw;
  int64_t (*pair_dwords)(int, int, int);
  int64_t (*written_dwords)(int, int, int);
};
template <int S>
PackInfo pack_info_of() {
  return PackInfo{Pack<S>::LO, Rec<S, 9>::RECDW, &Pack<S>::pair_dwords, &Pack<S>::written_dwords};
}
PackInfo pack_info(int S) { return S == 1 ? pack_info_of<1>() : (S == 2 ? pack_info_of<2>() : pack_info_of<3>()); }

// diet: the eight-wave form of the s=2 affine kernel (fill_affine_kernel, DIET): half-length ghost blocks,
// molecule A's codes not staged
size_t lds_need(int S, int NL, int team, int k1, int k2, int n, int m, bool dense = false, bool diet = false) {
  const int W = 2 * S + 1, PADB = S + 1;
  const size_t nv = (NL == 9 ? 12 : 1) * W;
  const size_t npad = diet ? 0 : (n + 3) & ~3, mpad = (m + 2 * PADB + 3) & ~3;
  const int nd = NL * W, np = nd / 4 + (nd % 4 ? 1 : 0), blk = diet ? 2 : ghost_blk(S);
  const size_t ring_dw = 2 * (((size_t)blk * W * np + 63) / 64 * 64) * 4;  // GhostFeed<S,NL>::RING_DW
  const size_t shared_dw = 16 + (size_t)k1 * k1 + (size_t)k2 * k2;  // progress words + score tables
  const size_t mu2_ring_dw = dense ? 2 * (size_t)blk * 64 : 0;  // Mu2Feed<S>::RING_DW
  return (team * (ring_dw + nv * NCOL + mu2_ring_dw) + shared_dw) * 4 + 2 * npad + 2 * mpad;
}

// fill_affine_slim_kernel (bialign_fill_slim.hpp), a workgroup of twelve waves: twelve ghost rings, a block of sentinels,
// progress words, score tables (lds_need_slim_base); per pair of the workgroup both molecules' codes (lds_need_slim_codes)
size_t lds_need_slim_base(int S, int k1, int k2) {
  const int W = 2 * S + 1;
  const int nd = 9 * W, np = nd / 4 + (nd % 4 ? 1 : 0), blk = ghost_blk(S);
  const size_t ring_dw = 2 * (((size_t)blk * W * np + 63) / 64 * 64) * 4;  // GhostFeed<S,9>::RING_DW
  return (12 * ring_dw + 4 * np + 16 + (size_t)k1 * k1 + (size_t)k2 * k2) * 4;
}
size_t lds_need_slim_codes(int S, int n, int m) {
  const int PADB = S + 1;
  const size_t npad = (n + 3) & ~3, mpad = (m + 2 * PADB + 3) & ~3;
  return 2 * npad + 2 * mpad;
}

int launch_fill(bialign_batch* b, const DeviceBatch& v, int first, int count) {
  if (b->wide) return launch_fill_wide(b, v, first, count);
  if (b->affine) {
    switch (b->S) {
      case 0: return launch_fill_affine<0>(b, v, first, count);
      case 1: return launch_fill_affine<1>(b, v, first, count);
      case 2: return launch_fill_affine<2>(b, v, first, count);
      case 3: return launch_fill_affine<3>(b, v, first, count);
      case 4: return launch_fill_affine<4>(b, v, first, count);
      case 5: return launch_fill_affine<5>(b, v, first, count);
    }
  } else {
    switch (b->S) {
      case 0: return launch_fill_linear<0>(b, v, first, count);
      case 1: return launch_fill_linear<1>(b, v, first, count);
      case 2: return launch_fill_linear<2>(b, v, first, count);
      case 3: return launch_fill_linear<3>(b, v, first, count);
      case 4: return launch_fill_linear<4>(b, v, first, count);
      case 5: return launch_fill_linear<5>(b, v, first, count);
    }
  }
  return fail(BIALIGN_E_UNSUPPORTED, "no fill kernel for affine=%d max_shift=%d", b->affine, b->S);
}

// Lean traceback of one chunk: as many (re-sweep, walk) rounds as its longest pair has strips.
int lean_traceback_rounds(bialign_batch* b, const DeviceBatch& v, int first, int count) {
  hipStream_t st = b->eng->stream;
  HIP_TRY(hipMemsetAsync(b->d_tstate.p, 0, sizeof(TraceState) * b->npairs, st));
  int rounds = 0;
  for (int t = first; t < first + count; ++t) rounds = std::max(rounds, b->pairs[b->order[t]].NS);
  rounds = (rounds + b->resw_k - 1) / b->resw_k;
  for (int r = 0; r < rounds; ++r) {
    int rc = BIALIGN_E_UNSUPPORTED;
#define BIALIGN_ROUND(S)                                                                              \
  case S:                                                                                             \
    rc = b->affine ? launch_resweep_affine<S>(b, v, first, count) : launch_resweep_linear<S>(b, v, first, count);          \
    if (rc == BIALIGN_OK)                                                                             \
      rc = b->affine ? launch_traceback_affine_strip<S>(b, v, first, count)                           \
                     : launch_traceback_linear_strip<S>(b, v, first, count);                          \
    break;
    switch (b->S) {
      BIALIGN_ROUND(0) BIALIGN_ROUND(1) BIALIGN_ROUND(2) BIALIGN_ROUND(3) BIALIGN_ROUND(4) BIALIGN_ROUND(5)
    }
#undef BIALIGN_ROUND
    if (rc) return rc;
  }
  return BIALIGN_OK;
}

int launch_traceback(const bialign_batch* b, const DeviceBatch& v, int first, int count, bool do_trace) {
  if (b->wide) return launch_traceback_wide(b, v, first, count, do_trace);
  if (b->affine) {
    switch (b->S) {
      case 0: return launch_traceback_affine<0>(b, v, first, count, do_trace);
      case 1: return launch_traceback_affine<1>(b, v, first, count, do_trace);
      case 2: return launch_traceback_affine<2>(b, v, first, count, do_trace);
      case 3: return launch_traceback_affine<3>(b, v, first, count, do_trace);
      case 4: return launch_traceback_affine<4>(b, v, first, count, do_trace);
      case 5: return launch_traceback_affine<5>(b, v, first, count, do_trace);
    }
  } else {
    switch (b->S) {
      case 0: return launch_traceback_linear<0>(b, v, first, count, do_trace);
      case 1: return launch_traceback_linear<1>(b, v, first, count, do_trace);
      case 2: return launch_traceback_linear<2>(b, v, first, count, do_trace);
      case 3: return launch_traceback_linear<3>(b, v, first, count, do_trace);
      case 4: return launch_traceback_linear<4>(b, v, first, count, do_trace);
      case 5: return launch_traceback_linear<5>(b, v, first, count, do_trace);
    }
  }
  return fail(BIALIGN_E_UNSUPPORTED, "no traceback kernel for affine=%d max_shift=%d", b->affine, b->S);
}

int launch_dump_any(const bialign_batch* b, const DeviceBatch& v, int pid, int32_t* d_out) {
  if (b->wide) return launch_dump_wide(b, v, pid, d_out);
  if (b->affine) {
    switch (b->S) {
      case 0: return launch_dump<0, 9>(b, v, pid, d_out);
      case 1: return launch_dump<1, 9>(b, v, pid, d_out);
      case 2: return launch_dump<2, 9>(b, v, pid, d_out);
      case 3: return launch_dump<3, 9>(b, v, pid, d_out);
      case 4: return launch_dump<4, 9>(b, v, pid, d_out);
      case 5: return launch_dump<5, 9>(b, v, pid, d_out);
    }
  } else {
    switch (b->S) {
      case 0: return launch_dump<0, 1>(b, v, pid, d_out);
      case 1: return launch_dump<1, 1>(b, v, pid, d_out);
      case 2: return launch_dump<2, 1>(b, v, pid, d_out);
      case 3: return launch_dump<3, 1>(b, v, pid, d_out);
      case 4: return launch_dump<4, 1>(b, v, pid, d_out);
      case 5: return launch_dump<5, 1>(b, v, pid, d_out);
    }
  }
  return fail(BIALIGN_E_UNSUPPORTED, "no dump kernel for affine=%d max_shift=%d", b->affine, b->S);
}

// Cut the batch into chunks of at most budget_dw dwords of layer storage and lay the pairs of each chunk end to
// end: as few chunks as the budget allows, of about equal size (an undersized last chunk would leave SIMDs
// idle); inside a chunk the longest sweeps are launched first.
int plan_chunks(bialign_batch* b, const std::vector<int64_t>& pair_dwords, int64_t budget_dw) {
  const int npairs = b->npairs;
  b->order.resize(npairs);
  std::iota(b->order.begin(), b->order.end(), 0);
  b->chunk_begin.assign(1, 0);
  b->max_chunk_dwords = 0;
  int64_t total_dw = 0;
  for (int p = 0; p < npairs; ++p) {
    if (pair_dwords[p] > budget_dw)
      return fail(BIALIGN_E_NOMEM, "pair %d needs %lld bytes of layers, budget is %lld", p,
                  (long long)pair_dwords[p] * 4, (long long)budget_dw * 4);
    total_dw += pair_dwords[p];
  }
  const int64_t want_chunks = (total_dw + budget_dw - 1) / budget_dw;
  const int64_t target_dw = std::min(budget_dw, (total_dw + want_chunks - 1) / want_chunks);
  int64_t used = 0;
  for (int p = 0; p < npairs; ++p) {
    if (used > 0 && (used + pair_dwords[p] > budget_dw || used >= target_dw)) {
      b->chunk_begin.push_back(p);
      used = 0;
    }
    b->pairs[p].scratch_off += used - b->pairs[p].layer_off;  // (relative to the pair's start until the first plan)
    b->pairs[p].layer_off = used;
    used += pair_dwords[p];
    b->max_chunk_dwords = std::max(b->max_chunk_dwords, used);
  }
  b->chunk_begin.push_back(npairs);
  for (size_t c = 0; c + 1 < b->chunk_begin.size(); ++c)
    std::stable_sort(b->order.begin() + b->chunk_begin[c], b->order.begin() + b->chunk_begin[c + 1],
                     [&](int x, int y) {
                       return b->wide ? b->pairs[x].n + b->pairs[x].m > b->pairs[y].n + b->pairs[y].m  // levels
                                      : b->pairs[x].G > b->pairs[y].G;
                     });
  return BIALIGN_OK;
}

// A batch laid out for packed records has to continue with full ones (an offset did not fit): cut it into chunks
// again, now by the pairs' full-record sizes, within the layer buffer it already holds (a larger one only if a
// single pair needs it), and hand the new layout to the device.
int replan_full(bialign_batch* b) {
  if (!b->packed_sizing) return BIALIGN_OK;
  b->packed_sizing = false;
  hipStream_t st = b->eng->stream;
  HIP_TRY(hipStreamSynchronize(st));
  const int64_t need = *std::max_element(b->full_dwords.begin(), b->full_dwords.end());
  if ((int64_t)b->d_layers.n < need + 16) HIP_TRY(b->d_layers.alloc((size_t)need + 16));
  if (int rc = plan_chunks(b, b->full_dwords, (int64_t)b->d_layers.n - 16)) return rc;
  HIP_TRY(hipMemcpy(b->d_pairs.p, b->pairs.data(), b->pairs.size() * sizeof(PairDesc), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(b->d_order.p, b->order.data(), b->order.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  return BIALIGN_OK;
}

int check_device_error(const bialign_batch* b) {
  int32_t err = 0;
  HIP_TRY(hipMemcpy(&err, b->d_err.p, sizeof err, hipMemcpyDeviceToHost));
  if (err) return fail(BIALIGN_E_DEVICE, "fill kernel: device error flag %d (1 = team hand-off timed out)", err);
  return BIALIGN_OK;
}

}  // namespace

extern "C" {

int bialign_abi_version(void) { return BIALIGN_ABI_VERSION; }

int bialign_build_experiment(void) { return BIALIGN_EXP; }

const char* bialign_last_error(void) { return g_err.c_str(); }

int bialign_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return fail(BIALIGN_E_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
  return n;
}

int bialign_engine_create(int device, bialign_engine** out) {
  if (!out) return fail(BIALIGN_E_INVALID, "out is NULL");
  *out = nullptr;
  int n = 0;
  HIP_TRY(hipGetDeviceCount(&n));
  if (device < 0 || device >= n) return fail(BIALIGN_E_INVALID, "device %d out of range (0..%d)", device, n - 1);
  HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(BIALIGN_E_UNSUPPORTED, "device %d is %s; this engine is built for gfx950 only", device,
                prop.gcnArchName);
  auto* e = new bialign_engine();
  e->device = device;
  e->num_cu = prop.multiProcessorCount;
  hipError_t err = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
  if (err == hipSuccess) err = hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking);
  for (int i = 0; i < 4 && err == hipSuccess; ++i) err = hipEventCreate(&e->ev[i]);
  if (err != hipSuccess) {
    bialign_engine_destroy(e);
    return fail(BIALIGN_E_DEVICE, "engine setup: %s", hipGetErrorString(err));
  }
  *out = e;
  return BIALIGN_OK;
}

void bialign_engine_destroy(bialign_engine* e) {
  if (!e) return;
  if (e->live_batches > 0) {  // destroy order is the caller's business (garbage collectors pick any): the
    e->closing = true;        // engine goes when its last batch goes
    return;
  }
  (void)hipSetDevice(e->device);
  for (auto& ev : e->ev)
    if (ev) (void)hipEventDestroy(ev);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
  delete e;
}

int bialign_batch_create(bialign_engine* eng, const bialign_params* prm, const bialign_scoring* sc,
                         const bialign_pairs* pr, int64_t hbm_budget, bialign_batch** out) {
  if (!eng || !prm || !sc || !pr || !out) return fail(BIALIGN_E_INVALID, "NULL argument");
  *out = nullptr;
  if (pr->npairs < 1) return fail(BIALIGN_E_INVALID, "npairs must be >= 1");
  if (prm->max_shift < 0) return fail(BIALIGN_E_INVALID, "max_shift must be >= 0");
  if (prm->max_shift > BIALIGN_MAX_SHIFT)
    return fail(BIALIGN_E_UNSUPPORTED, "max_shift %d > %d", prm->max_shift, BIALIGN_MAX_SHIFT);
  if (sc->k1 < 1 || sc->k1 > 256 || sc->k2 < 1 || sc->k2 > 256 || !sc->s1 || !sc->s2)
    return fail(BIALIGN_E_INVALID, "scoring tables: k1,k2 must be 1..256 and tables non-NULL");
  HIP_TRY(hipSetDevice(eng->device));

  std::unique_ptr<bialign_batch> b(new bialign_batch());
  b->eng = eng;
  b->prm = *prm;
  if (prm->recurrence < BIALIGN_REC_AUTO || prm->recurrence > BIALIGN_REC_LINEAR)
    return fail(BIALIGN_E_INVALID, "recurrence must be 0 (auto), 1 (affine) or 2 (non-affine)");
  b->affine = prm->recurrence == BIALIGN_REC_AUTO ? prm->gap_opening_cost != 0  // pyx:204-205, 444
                                                  : prm->recurrence == BIALIGN_REC_AFFINE;
  b->NL = b->affine ? 9 : 1;
  b->S = prm->max_shift;
  b->npairs = pr->npairs;
  b->k1 = sc->k1;
  b->k2 = sc->k2;
  b->dense = pr->mu2_dense != nullptr;
  b->lean_trace = (prm->flags & BIALIGN_BATCH_LEAN_TRACE) != 0;
  b->lean = b->lean_trace || (prm->flags & BIALIGN_BATCH_SCORE_ONLY) != 0;
  b->wide = prm->max_shift > BIALIGN_MAX_SHIFT_TILED;  // bialign_wide.hpp: anti-diagonal path, all layers in HBM
  // wide bands: score-only batches of the affine recurrence keep just the ring of derived values (bialign_wide.hpp);
  // the memory-lean traceback and the one-layer recurrence's score-only form exist for the tiled sweeps only
  if (b->wide && b->lean && (b->lean_trace || !b->affine))
    return fail(BIALIGN_E_UNSUPPORTED, "LEAN_TRACE, and SCORE_ONLY of the non-affine recurrence, exist for max_shift <= %d only",
                BIALIGN_MAX_SHIFT_TILED);
  if (b->dense && !pr->mu2_off) return fail(BIALIGN_E_INVALID, "mu2_dense given without mu2_off");
  if (!b->dense && (!pr->cls_a || !pr->cls_b)) return fail(BIALIGN_E_INVALID, "cls_a / cls_b are NULL (LOOKUP form)");
  const int S = b->S, W = 2 * S + 1;

  // int32 safety window: finite scores and the drift of "-infinity" cells must
  // stay within 2^28 of where they start (kernels rely on it, see THRESH).
  int64_t amax = 0;
  for (int t = 0; t < sc->k1 * sc->k1; ++t) amax = std::max<int64_t>(amax, std::llabs((long long)sc->s1[t]));
  int64_t bmax = 0;
  for (int t = 0; t < sc->k2 * sc->k2; ++t) bmax = std::max<int64_t>(bmax, std::llabs((long long)sc->s2[t]));
  if (b->dense) {  // dense mu2: the bound comes from the tables themselves
    bmax = 0;
    for (int p = 0; p < pr->npairs; ++p) {
      const int64_t cnt = (int64_t)std::max(pr->len_a[p], 0) * std::max(pr->len_b[p], 0);
      for (int64_t t = 0; t < cnt; ++t)
        bmax = std::max<int64_t>(bmax, std::llabs((long long)pr->mu2_dense[pr->mu2_off[p] + t]));
    }
  }
  const int64_t colmax = amax + bmax + 2 * (std::llabs((long long)prm->gap_cost) + std::llabs((long long)prm->gap_opening_cost)) +
                         2 * std::llabs((long long)prm->shift_cost);

  int64_t tot_a = 0, tot_b = 0, tot_mu2 = 0;
  b->pairs.resize(pr->npairs);
  std::vector<int64_t> pair_dwords(pr->npairs);
  for (int p = 0; p < pr->npairs; ++p) {
    const int n = pr->len_a[p], m = pr->len_b[p];
    if (n < 1 || m < 1)  // the reference raises IndexError on empty molecules (pyx:407)
      return fail(BIALIGN_E_INVALID, "pair %d: empty molecule (n=%d, m=%d)", p, n, m);
    if ((2 * ((int64_t)n + m) + 8) * colmax >= (1 << 28))
      return fail(BIALIGN_E_RANGE, "pair %d: scores may leave the int32 safety window (n+m=%d, column bound %lld)", p,
                  n + m, (long long)colmax);
    PairDesc& d = b->pairs[p];
    d.n = n;
    d.m = m;
    d.NS = d.P = d.G = 0;
    if (!b->wide) sweep_geometry(n, m, S, &d.NS, &d.P, &d.G);
    d.trace_cap = 2 * (n + m) + 2;
    d.seq_a = pr->off_a[p];
    d.seq_b = pr->off_b[p];
    d.trace_off = b->trace_bytes;
    d.mu2_off = b->dense ? pr->mu2_off[p] : 0;
    if (b->dense) tot_mu2 = std::max<int64_t>(tot_mu2, pr->mu2_off[p] + (int64_t)n * m);
    b->trace_bytes += d.trace_cap;
    b->cells += cells_of(n, m, S);
    tot_a = std::max<int64_t>(tot_a, pr->off_a[p] + n);
    tot_b = std::max<int64_t>(tot_b, pr->off_b[p] + m);
    if (!b->wide) {
      b->lds_bytes = std::max(b->lds_bytes, lds_need(S, b->NL, 1, sc->k1, sc->k2, n, m, b->dense));
      b->lds_base = std::max(b->lds_base, lds_need(S, b->NL, 0, sc->k1, sc->k2, n, m, b->dense));
      b->lds_diet8 = std::max(b->lds_diet8, lds_need(S, b->NL, 8, sc->k1, sc->k2, n, m, false, true));
      b->lds_slim_codes = std::max(b->lds_slim_codes, lds_need_slim_codes(S, n, m));
    }
    b->lds_trace = std::max<size_t>(b->lds_trace, ((size_t)sc->k1 * sc->k1 + (size_t)sc->k2 * sc->k2) * 4 +
                                                      2 * (size_t)((n + 3) & ~3) + 2 * (size_t)((m + 3) & ~3));
  }
  if (!b->wide)
    b->lds_per_wave = lds_need(S, b->NL, 1, sc->k1, sc->k2, 1, 1, b->dense) - lds_need(S, b->NL, 0, sc->k1, sc->k2, 1, 1, b->dense);
  if (!b->wide) b->lds_slim_base = lds_need_slim_base(S, sc->k1, sc->k2);
  if (std::max(b->lds_bytes, b->lds_trace) > 160 * 1024)
    return fail(BIALIGN_E_UNSUPPORTED, "molecules too long for the LDS staging (%zu bytes needed, 160 KiB per workgroup)",
                std::max(b->lds_bytes, b->lds_trace));

  // ---- packed records (Pack<S>): for sweeps whose steps are mostly interior
  {
    const char* e = getenv("BIALIGN_PACK");  // "0" never, "1" wherever the layout allows (tests), unset: when it pays
    const bool force = e && e[0] == '1';
    const PackInfo pki = pack_info(S);
    bool ok = b->affine && S >= 1 && S <= BIALIGN_MAX_SHIFT_PACKED && !b->lean && prm->gap_opening_cost <= 0 && !(e && e[0] == '0') &&
              (force || colmax < 8192) &&  // offsets span a few column scores (measured: up to 2.5): beyond this they will not fit
              // s=3 runs one wave per SIMD and is bound by issue: packing pays where the device is full (512 pairs x len 512
              // +7 %, 86 pairs in cross-CU teams of 11 +25 %), not for a few long pairs (21 x len 1024: -14 %, 8 x len 2048: -15 %)
              (force || S < 3 || pr->npairs >= 64);
    for (int p = 0; ok && p < pr->npairs; ++p) {
      const PairDesc& d = b->pairs[p];
      const int interior = d.m - S - pki.lo + 1;  // phases LO .. m - S per strip
      const int64_t packed_dw = pki.written_dwords(d.G, d.P, d.m);
      const int64_t full_dw = (int64_t)d.G * pki.full_recdw;
      // unless forced (tests): only where it saves a fifth of the bytes written (long enough columns, more than a strip or two)
      ok = interior >= 1 && (force || packed_dw * 5 <= full_dw * 4);
    }
    b->pack = ok;
  }

  // ---- chunking under the HBM budget; inside a chunk longest sweeps first
  size_t free_b = 0, total_b = 0;
  HIP_TRY(hipMemGetInfo(&free_b, &total_b));
  free_b += (eng->layer_cache.n + eng->layer_cache2.n) * sizeof(int32_t);  // reused or released below, ours either way
  int64_t budget = hbm_budget > 0 ? hbm_budget : (int64_t)(free_b * 0.85);
  budget = std::min<int64_t>(budget, (int64_t)(free_b * 0.95));
  const int64_t budget_dw = budget / 4;
  // layer storage per pair in the batch's mode (dwords); a pair's scratch records follow its LEAN records
  auto size_pairs = [&]() {
    for (int p = 0; p < pr->npairs; ++p) {
      PairDesc& d = b->pairs[p];
      if (b->wide) {  // reference-order layers, every band slot of every (i, j); none at all for a score-only batch
        pair_dwords[p] = b->lean ? 16 : wide_pair_dwords(d.n, d.m, S, b->NL);
        continue;
      }
      int slp = (64 / W - 1) * W;  // Rec<S,NL>::SLP
      if ((slp + 7) / 8 * 8 - slp <= BIALIGN_PADMAX) slp = (slp + 7) / 8 * 8;
      const int64_t full_rec = (int64_t)((b->NL * W) / 4) * slp * 4 + 64 * ((b->NL * W) % 4);  // Rec<S,NL>::RECDW per step
      const int64_t lean_dw = (int64_t)d.G * ((W * b->NL * W + 3) / 4 * 4);       // Rec<S,NL,true>::RECDW per step
      const int64_t scratch_dw = (int64_t)(d.m + 2 * (64 / W - 1) + W) * full_rec;  // one strip: m + MAXOFF + 1 records
      d.scratch_off = lean_dw;  // relative to layer_off until the chunk layout is fixed below
      pair_dwords[p] = b->lean_trace ? lean_dw + b->resw_k * scratch_dw : (b->lean ? lean_dw : (int64_t)d.G * full_rec);
      if (b->pack && !b->lean)  // (a sweep that meets an unpackable value is repeated with full records: replan_full())
        pair_dwords[p] = pack_info(S).pair_dwords(d.G, d.P, d.m);
    }
  };
  // lean traceback: few pairs -> several strips per round (they re-sweep in parallel), as memory allows
  auto pick_resw_k = [&]() {
    // as many strips per round as keep ~2048 waves busy -- re-sweeps of different strips are independent, so a
    // single long pair gets up to 256 at once -- but no more scratch than about a quarter of the pair's full
    // layers (a strip's scratch is 1/NS of them): the mode exists to save memory
    int ns_max = 1;
    for (const PairDesc& d : b->pairs) ns_max = std::max(ns_max, d.NS);
    b->resw_k = (int)std::min<int64_t>(std::min<int64_t>(256, std::max(1, ns_max / 4)), std::max<int64_t>(1, 2048 / pr->npairs));
    if (const char* e = getenv("BIALIGN_RESW_K")) b->resw_k = std::min(256, std::max(1, atoi(e)));  // tests
    for (size_pairs(); b->resw_k > 1 && *std::max_element(pair_dwords.begin(), pair_dwords.end()) > budget_dw; size_pairs())
      b->resw_k /= 2;
  };
  if (b->lean_trace) pick_resw_k();
  size_pairs();
  // A pair whose full layers exceed the budget is served from reduced storage instead of failing
  // (memory-lean traceback, ~1.3x the time).
  if (b->pack) {  // the fallback to full records must be possible within the same budget
    int64_t full_max = 0;
    for (const PairDesc& d : b->pairs)
      full_max = std::max(full_max, (int64_t)d.G * pack_info(S).full_recdw);
    if (std::max(full_max, *std::max_element(pair_dwords.begin(), pair_dwords.end())) > budget_dw) {
      b->pack = false;
      size_pairs();
    }
  }
  if (!b->lean && !b->wide &&
      *std::max_element(pair_dwords.begin(), pair_dwords.end()) > budget_dw) {
    b->lean = b->lean_trace = true;
    b->pack = false;
    pick_resw_k();
  }
  b->full_dwords.resize(pr->npairs);
  for (int p = 0; p < pr->npairs; ++p) {
    int slp = (64 / W - 1) * W;  // as in size_pairs
    if ((slp + 7) / 8 * 8 - slp <= BIALIGN_PADMAX) slp = (slp + 7) / 8 * 8;
    b->full_dwords[p] = b->wide ? pair_dwords[p] : (int64_t)b->pairs[p].G * ((int64_t)((b->NL * W) / 4) * slp * 4 + 64 * ((b->NL * W) % 4));
  }
  b->packed_sizing = b->pack && !b->lean;
  if (int rc = plan_chunks(b.get(), pair_dwords, budget_dw)) return rc;

  // ---- the layer buffer: a cached one if large enough, else a new one; if the device cannot provide a chunk of the
  //      planned size after all (fragmentation, another tenant), plan smaller chunks and try again
  size_t layer_dw = 0;
  for (int attempt = 0;; ++attempt) {
    layer_dw = (size_t)b->max_chunk_dwords + 16;  // slack: ghost tail pieces are read 16 B wide
    DevBuf<int32_t>* slot = nullptr;  // the smallest cached buffer that is large enough
    for (DevBuf<int32_t>* c : {&eng->layer_cache, &eng->layer_cache2})
      if (c->p && c->n >= layer_dw && (!slot || c->n < slot->n)) slot = c;
    if (slot) {
      b->d_layers.swap(*slot);
      break;
    }
    eng->layer_cache.release();
    eng->layer_cache2.release();
    hipError_t err = b->d_layers.alloc(layer_dw);
    if (err == hipSuccess && attempt == 0 && getenv("BIALIGN_TEST_FAIL_ALLOC")) {  // tests: pretend the first allocation failed
      b->d_layers.release();
      err = hipErrorOutOfMemory;
    }
    if (err == hipSuccess) break;
    (void)hipGetLastError();
    b->d_layers.p = nullptr;
    b->d_layers.n = 0;
    const int64_t smaller = (int64_t)b->max_chunk_dwords * 3 / 4;
    if (attempt >= 3 || smaller < *std::max_element(pair_dwords.begin(), pair_dwords.end()))
      return fail(BIALIGN_E_DEVICE, "hipMalloc of %zu bytes of layer storage failed: %s", layer_dw * 4, hipGetErrorString(err));
    for (PairDesc& d : b->pairs) d.scratch_off -= d.layer_off, d.layer_off = 0;  // back to pair-relative, as before the first plan
    if (int rc = plan_chunks(b.get(), pair_dwords, smaller)) return rc;
  }

  // ---- upload (own stream: a batch can be prepared while another one sweeps)
  hipStream_t st = eng->copy_stream;
  HIP_TRY(b->d_pairs.upload(b->pairs.data(), b->pairs.size(), st));
  HIP_TRY(b->d_order.upload(b->order.data(), b->order.size(), st));
  HIP_TRY(b->d_s1.upload(sc->s1, (size_t)sc->k1 * sc->k1, st));
  HIP_TRY(b->d_s2.upload(sc->s2, (size_t)sc->k2 * sc->k2, st));
  HIP_TRY(b->d_seq_a.upload(pr->seq_a, tot_a, st));
  std::vector<uint8_t> zeros;
  if (b->dense) zeros.assign((size_t)std::max(tot_a, tot_b), 0);  // class codes are unused in dense mode
  HIP_TRY(b->d_cls_a.upload(b->dense ? zeros.data() : pr->cls_a, tot_a, st));
  HIP_TRY(b->d_seq_b.upload(pr->seq_b, tot_b, st));
  HIP_TRY(b->d_cls_b.upload(b->dense ? zeros.data() : pr->cls_b, tot_b, st));
  if (b->dense) HIP_TRY(b->d_mu2.upload(pr->mu2_dense, (size_t)tot_mu2, st));
  if (getenv("BIALIGN_DEBUG")) {  // placement study: address and plain streaming-write rate of the layer buffer
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      HIP_TRY(hipEventRecord(eng->ev[0], st));
      HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)b->d_layers.p, 0, layer_dw, st));
      HIP_TRY(hipEventRecord(eng->ev[1], st));
      HIP_TRY(hipEventSynchronize(eng->ev[1]));
      HIP_TRY(hipEventElapsedTime(&ms, eng->ev[0], eng->ev[1]));
    }
    fprintf(stderr, "[bialign] layers %p (%.1f GiB) memset %.0f GB/s\n", (void*)b->d_layers.p,
            layer_dw * 4.0 / (1 << 30), layer_dw * 4.0 / ms / 1e6);
  }
  HIP_TRY(b->d_scores.alloc(pr->npairs));
  if (b->lean_trace) HIP_TRY(b->d_tstate.alloc(pr->npairs));
  HIP_TRY(b->d_tlen.alloc(pr->npairs));
  HIP_TRY(b->d_complete.alloc(pr->npairs));
  HIP_TRY(b->d_err.alloc(1));
  HIP_TRY(hipMemsetAsync(b->d_err.p, 0, sizeof(int32_t), st));
  if (const char* e = getenv("BIALIGN_XCU_SPIN_LIMIT")) b->xcu_spin_limit = std::max(0, atoi(e));  // tests: force hand-off timeouts
  HIP_TRY(b->d_trace.alloc(b->trace_bytes));
  HIP_TRY(hipMemsetAsync(b->d_tlen.p, 0, sizeof(int32_t) * pr->npairs, st));
  HIP_TRY(hipMemsetAsync(b->d_complete.p, 0, sizeof(int32_t) * pr->npairs, st));
  HIP_TRY(hipEventCreateWithFlags(&b->uploaded, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(b->uploaded, st));
  HIP_TRY(hipStreamSynchronize(st));  // the caller's host arrays may go away now
  ++eng->live_batches;
  *out = b.release();
  return BIALIGN_OK;
}

void bialign_batch_destroy(bialign_batch* b) {
  if (!b) return;
  (void)hipSetDevice(b->eng->device);
  bialign_engine* eng = b->eng;
  if (b->pending) (void)hipStreamSynchronize(b->eng->stream);  // its kernels still use the buffers freed below
  if (b->d_layers.p && !eng->closing) {  // keep the buffer for the next batch: a free slot, else in place of a smaller one
    (void)hipStreamSynchronize(eng->stream);
    DevBuf<int32_t>* slot = !eng->layer_cache.p ? &eng->layer_cache : (!eng->layer_cache2.p ? &eng->layer_cache2 : nullptr);
    if (!slot) slot = eng->layer_cache.n <= eng->layer_cache2.n ? &eng->layer_cache : &eng->layer_cache2;
    if (!slot->p || b->d_layers.n > slot->n) slot->swap(b->d_layers);
  }
  delete b;
  if (--eng->live_batches == 0 && eng->closing) bialign_engine_destroy(eng);
}

// streaming-write rate of a buffer in GB/s (second of two memset passes)
static int probe_write_rate(bialign_engine* e, int32_t* p, size_t dwords, double* gbps) {
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    HIP_TRY(hipEventRecord(e->ev[0], e->stream));
    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)p, 0, dwords, e->stream));
    HIP_TRY(hipEventRecord(e->ev[1], e->stream));
    HIP_TRY(hipEventSynchronize(e->ev[1]));
    HIP_TRY(hipEventElapsedTime(&ms, e->ev[0], e->ev[1]));
  }
  *gbps = dwords * 4.0 / (ms * 1e6);
  return BIALIGN_OK;
}

int bialign_engine_reserve(bialign_engine* e, int64_t bytes, int tries, double* rate_gbps) {
  if (!e || bytes <= 0) return fail(BIALIGN_E_INVALID, "bad argument");
  HIP_TRY(hipSetDevice(e->device));
  const size_t dwords = ((size_t)bytes + 3) / 4;
  DevBuf<int32_t> best;
  double best_rate = 0;
  if (e->layer_cache.n >= dwords) {  // what is cached is the first candidate
    best.swap(e->layer_cache);
  } else {
    e->layer_cache.release();
    HIP_TRY(best.alloc(dwords));
  }
  int rc = probe_write_rate(e, best.p, dwords, &best_rate);
  for (int t = 1; t < tries && rc == BIALIGN_OK; ++t) {
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    if (free_b < (size_t)(dwords * 4 * 1.05)) break;  // no room for a second candidate next to the held one
    DevBuf<int32_t> cand;
    if (cand.alloc(dwords) != hipSuccess) { (void)hipGetLastError(); break; }
    double rate = 0;
    rc = probe_write_rate(e, cand.p, dwords, &rate);
    if (rc == BIALIGN_OK && rate > best_rate * 1.005) {  // keep the better one; the other goes back
      best.swap(cand);
      best_rate = rate;
    }
  }
  if (rc == BIALIGN_OK) {
    e->layer_cache.swap(best);
    if (rate_gbps) *rate_gbps = best_rate;
  }
  return rc;
}

int bialign_engine_trim(bialign_engine* e) {
  if (!e) return fail(BIALIGN_E_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(e->device));
  e->layer_cache.release();
  e->layer_cache2.release();
  return BIALIGN_OK;
}

int bialign_batch_get_info(const bialign_batch* b, bialign_batch_info* info) {
  if (!b || !info) return fail(BIALIGN_E_INVALID, "NULL argument");
  info->npairs = b->npairs;
  info->nchunks = (int)b->chunk_begin.size() - 1;
  info->affine = b->affine;
  info->max_shift = b->S;
  info->cells = b->cells;
  info->layer_bytes = b->cells * 4 * b->NL;
  info->hbm_layer_bytes = b->max_chunk_dwords * 4;
  info->trace_bytes = b->trace_bytes;
  info->storage = b->lean_trace ? BIALIGN_BATCH_LEAN_TRACE : (b->lean ? BIALIGN_BATCH_SCORE_ONLY : 0);
  info->reserved = 0;
  return BIALIGN_OK;
}

// Enqueue one run of the batch on the engine's stream (all chunks: fill, then traceback).
static int enqueue_run(bialign_batch* b, uint32_t flags) {
  HIP_TRY(hipSetDevice(b->eng->device));
  const bool do_trace = !(flags & BIALIGN_RUN_FILL_ONLY) && (!b->lean || b->lean_trace);
  const DeviceBatch v = b->view();
  hipStream_t st = b->eng->stream;
  b->timing = bialign_timing{};
  b->ran = b->ran_trace = false;
  b->used_xcu = b->used_pack = false;
  const int nchunks = (int)b->chunk_begin.size() - 1;
  while ((int)b->evs.size() < 3 * nchunks) {
    hipEvent_t e = nullptr;
    HIP_TRY(hipEventCreate(&e));
    b->evs.push_back(e);
  }
  HIP_TRY(hipStreamWaitEvent(st, b->uploaded, 0));
  HIP_TRY(hipMemsetAsync(b->d_err.p, 0, sizeof(int32_t), st));  // the flag is per run
  for (int c = 0; c < nchunks; ++c) {  // stream order keeps chunk c's traceback ahead of chunk c+1's sweep
    const int first = b->chunk_begin[c], count = b->chunk_begin[c + 1] - first;
    HIP_TRY(hipEventRecord(b->evs[3 * c], st));
    int rc = launch_fill(b, v, first, count);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(b->evs[3 * c + 1], st));
    if (!b->lean) {  // (with LEAN records the sweep itself wrote the scores)
      rc = launch_traceback(b, v, first, count, do_trace);
      if (rc) return rc;
    } else if (b->lean_trace && do_trace) {
      rc = lean_traceback_rounds(b, v, first, count);
      if (rc) return rc;
    }
    HIP_TRY(hipEventRecord(b->evs[3 * c + 2], st));
    b->timing.fill_launches += 1;
    b->timing.traceback_launches += 1;
    b->timing.waves_per_pair = std::abs(b->last_team);
    b->timing.cross_cu = b->last_team < 0;
  }
  b->pending = true;
  b->pending_trace = do_trace;
  b->pending_flags = flags;
  return BIALIGN_OK;
}

int bialign_batch_wait(bialign_batch* b) {
  if (!b) return fail(BIALIGN_E_INVALID, "NULL batch");
  if (!b->pending) return BIALIGN_OK;
  HIP_TRY(hipSetDevice(b->eng->device));
  for (;;) {
    b->pending = false;
    const int nchunks = (int)b->chunk_begin.size() - 1;
    HIP_TRY(hipEventSynchronize(b->evs[3 * nchunks - 1]));  // (a re-planned batch may have fewer chunks than events)
    for (int c = 0; c < nchunks; ++c) {
      float f = 0, t = 0;
      HIP_TRY(hipEventElapsedTime(&f, b->evs[3 * c], b->evs[3 * c + 1]));
      HIP_TRY(hipEventElapsedTime(&t, b->evs[3 * c + 1], b->evs[3 * c + 2]));
      b->timing.fill_ms += f;
      b->timing.traceback_ms += t;
    }
    int32_t err = 0;
    HIP_TRY(hipMemcpy(&err, b->d_err.p, sizeof err, hipMemcpyDeviceToHost));
    if (!err) break;
    // Bit 1: a cross-CU team lost co-residency (its waves spin on partners that were never scheduled:
    // another tenant holds wave slots) -- the run is repeated with in-workgroup teams, which depend on
    // nobody.  Bit 2: a packed record met a value that does not fit its 16-bit offset -- the run is
    // repeated with full records.  Either way the batch stays on the safe form.
    bool again = false;
    if (err & 1) {
      if (!b->used_xcu || b->no_xcu)
        return fail(BIALIGN_E_DEVICE, "fill kernel: team hand-off timed out (device error flag %d)", err);
      b->no_xcu = again = true;
    }
    if (err & 2) {
      if (!b->used_pack || b->pack_failed)
        return fail(BIALIGN_E_DEVICE, "fill kernel: device error flag %d", err);
      b->pack_failed = again = true;
      if (int rc = replan_full(b)) return rc;
    }
    if (!again) return fail(BIALIGN_E_DEVICE, "fill kernel: device error flag %d", err);
    ++b->recovered;
    if (int rc = enqueue_run(b, b->pending_flags)) return rc;
  }
  b->timing.recovered_runs = b->recovered;
  b->timing.packed_records = b->used_pack ? 1 : 0;
  b->ran = true;
  b->ran_trace = b->pending_trace;
  return BIALIGN_OK;
}

int bialign_batch_run(bialign_batch* b, uint32_t flags) {
  if (!b) return fail(BIALIGN_E_INVALID, "NULL batch");
  int rc = bialign_batch_wait(b);  // one run of a batch at a time
  if (rc) return rc;
  rc = enqueue_run(b, flags);
  if (rc) return rc;
  return (flags & BIALIGN_RUN_ASYNC) ? BIALIGN_OK : bialign_batch_wait(b);
}

int bialign_batch_get_timing(const bialign_batch* b, bialign_timing* t) {
  if (!b || !t) return fail(BIALIGN_E_INVALID, "NULL argument");
  if (int rc = bialign_batch_wait(const_cast<bialign_batch*>(b))) return rc;
  *t = b->timing;
  return BIALIGN_OK;
}

int bialign_batch_get_scores(const bialign_batch* b, int32_t* scores) {
  if (!b || !scores) return fail(BIALIGN_E_INVALID, "NULL argument");
  if (int rc = bialign_batch_wait(const_cast<bialign_batch*>(b))) return rc;
  if (!b->ran) return fail(BIALIGN_E_INVALID, "bialign_batch_run has not been called");
  HIP_TRY(hipSetDevice(b->eng->device));
  HIP_TRY(hipMemcpy(scores, b->d_scores.p, sizeof(int32_t) * b->npairs, hipMemcpyDeviceToHost));
  return BIALIGN_OK;
}

int bialign_batch_get_traces(const bialign_batch* b, uint8_t* trace, int64_t* trace_off, int32_t* trace_len,
                             int32_t* complete) {
  if (!b || !trace || !trace_off || !trace_len || !complete) return fail(BIALIGN_E_INVALID, "NULL argument");
  if (int rc = bialign_batch_wait(const_cast<bialign_batch*>(b))) return rc;
  if (b->lean && !b->lean_trace)
    return fail(BIALIGN_E_INVALID, "batch was created with BIALIGN_BATCH_SCORE_ONLY: it holds no layers to trace back");
  if (!b->ran || !b->ran_trace) return fail(BIALIGN_E_INVALID, "no traceback has been run on this batch");
  HIP_TRY(hipSetDevice(b->eng->device));
  HIP_TRY(hipMemcpy(trace, b->d_trace.p, b->trace_bytes, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(trace_len, b->d_tlen.p, sizeof(int32_t) * b->npairs, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(complete, b->d_complete.p, sizeof(int32_t) * b->npairs, hipMemcpyDeviceToHost));
  for (int p = 0; p < b->npairs; ++p) trace_off[p] = b->pairs[p].trace_off;
  return BIALIGN_OK;
}

int bialign_batch_dump_layers(bialign_batch* b, int32_t pair, int32_t* out) {
  if (!b || !out) return fail(BIALIGN_E_INVALID, "NULL argument");
  if (pair < 0 || pair >= b->npairs) return fail(BIALIGN_E_INVALID, "pair %d out of range", pair);
  if (int rc = bialign_batch_wait(b)) return rc;
  if (b->lean) return fail(BIALIGN_E_INVALID, "batch was created with reduced layer storage (SCORE_ONLY / LEAN_TRACE): it holds no full layers");
  HIP_TRY(hipSetDevice(b->eng->device));
  hipStream_t st = b->eng->stream;
  // one-pair launch out of the regular launch order (team shape and layer offset are the pair's own)
  int pos = (int)(std::find(b->order.begin(), b->order.end(), pair) - b->order.begin());
  DeviceBatch v = b->view();
  HIP_TRY(hipMemsetAsync(b->d_err.p, 0, sizeof(int32_t), st));
  b->used_xcu = b->used_pack = false;
  int rc = launch_fill(b, v, pos, 1);
  if (rc) return rc;
  if ((b->used_xcu && !b->no_xcu) || (b->used_pack && !b->pack_failed)) {  // forms a launch can fall back from, as in a run
    HIP_TRY(hipStreamSynchronize(st));
    int32_t err = 0;
    HIP_TRY(hipMemcpy(&err, b->d_err.p, sizeof err, hipMemcpyDeviceToHost));
    if (err) {
      if (err & 1) b->no_xcu = true;
      if (err & 2) {
        b->pack_failed = true;
        if (int rc2 = replan_full(b)) return rc2;
        // the re-plan sorts every chunk's launch order anew, moves every pair's layer_off and may have replaced
        // the layer buffer: the pair's launch position and the device view are the new ones from here on
        v = b->view();
        pos = (int)(std::find(b->order.begin(), b->order.end(), pair) - b->order.begin());
      }
      ++b->recovered;
      HIP_TRY(hipMemsetAsync(b->d_err.p, 0, sizeof(int32_t), st));
      rc = launch_fill(b, v, pos, 1);
      if (rc) return rc;
    }
  }
  const PairDesc& d = b->pairs[pair];
  const int W = 2 * b->S + 1;
  const size_t elems = (size_t)b->NL * (d.n + 1) * (d.m + 1) * W * W;
  DevBuf<int32_t> d_out;
  HIP_TRY(d_out.alloc(elems));
  rc = launch_dump_any(b, v, pair, d_out.p);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(out, d_out.p, elems * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  // scores and traces live in their own buffers and stay valid; only the layer region was rewritten
  return check_device_error(b);
}

}  // extern "C"
