// abrk_kernels.h - __global__ kernels (one lane = one arm instance) and the per-arm
// launch table.  Each arm translation unit (abrk_arm_<name>.hip, abrk_arm_rt<N>.hip)
// instantiates these for its arm policy and both arithmetic types.
#pragma once
#include <cstdlib>

#include "abrk_rows.h"

namespace abrk {

constexpr int kBlock = 64;  // rows are independent, no LDS sharing: one wavefront per workgroup
constexpr int kMinWaves = 1;  // kernels without a register cap: whatever occupancy their register count allows
// Measurement switches of the host-side launch logic (ABRK_NO_HANDOVER, ABRK_FINISH_SLOTS, ABRK_OBS_PLAIN, ...: listed in
// INTEGRATION.md) are read only when ABRK_MEASUREMENT=1 is set: no stray environment variable changes which algorithm a
// production call runs.
inline const char* measurement_env(const char* name) {
  static const bool on = [] {
    const char* e = getenv("ABRK_MEASUREMENT");
    return e && e[0] == '1';
  }();
  return on ? getenv(name) : nullptr;
}

#define ABRK_ROW_INDEX                                     \
  long b = (long)blockIdx.x * kBlock + threadIdx.x;        \
  if (b >= B) return;

// Cooperative, coalesced row stores: each lane parks its row in the wavefront's LDS slab, laid out exactly
// like the 64-row block of the output array, then the 64 lanes stream the slab out linearly with 16-byte
// stores (1 KiB contiguous per instruction, no index arithmetic).  The row-per-lane alternative writes 64
// separate 8-byte pieces per instruction and is what makes the full-output mode HBM-inefficient.  (The
// parking writes have a 2-4-way bank conflict for even row lengths; LDS time is hidden behind HBM here.)
// widest output row of an N-joint arm: J/dJ (6N), M/C (N*N), T (16)
constexpr int max_row(int n) { return (6 * n > n * n ? 6 * n : n * n) > 16 ? (6 * n > n * n ? 6 * n : n * n) : 16; }
template <class T>
struct LdsStore {
  T* buf;
  long row0, B;
  int lane;
  template <int R>
  __device__ __forceinline__ void put(T* __restrict__ out, long, bool, const T (&v)[R]) {
    constexpr int V = 16 / sizeof(T);  // elements per 16-byte store
    sfor<R>([&](auto k) ABRK_LAMBDA { buf[lane * R + k()] = v[k()]; });
    __syncthreads();
    const long left = B - row0;
    const int total = (int)(left < kBlock ? left : (long)kBlock) * R;  // elements of this block
    T* o = out + row0 * R;                                            // 64*R*sizeof(T)-byte aligned
    const bool aligned = (reinterpret_cast<unsigned long long>(o) & 15ull) == 0;  // caller pointers may not be
    constexpr int iters = (kBlock * R / V + kBlock - 1) / kBlock;
    sfor<iters>([&](auto j) ABRK_LAMBDA {
      const int e = (lane + kBlock * j()) * V;
      if (aligned && e + V <= total) {
        using vec = T __attribute__((ext_vector_type(V)));
        *reinterpret_cast<vec*>(o + e) = *reinterpret_cast<const vec*>(buf + e);
      } else {
        sfor<V>([&](auto c) ABRK_LAMBDA {
          if (e + c() < total) o[e + c()] = buf[e + c()];
        });
      }
    });
    __syncthreads();
  }
};

template <class A, class T, bool WITH_DQ>
__global__ void __launch_bounds__(kBlock, kMinWaves)
dyn_kernel(A arm, int frame, int m, T ox, T oy, T oz, unsigned want, long B, const T* __restrict__ qg,
           const T* __restrict__ dqg, DynOutP<T> out) {
  __shared__ __attribute__((aligned(16))) T slab[kBlock * max_row(A::N)];
  const long row0 = (long)blockIdx.x * kBlock;
  const long b = row0 + threadIdx.x;
  LdsStore<T> st{slab, row0, B, (int)threadIdx.x};
  dyn_body<A, T, WITH_DQ>(b, b < B, st, arm, frame, m, ox, oy, oz, want, B, qg, dqg, out);
}

// Register budget: the law without Coriolis vector and without the rarely used optional inputs is
// (xyz kernel) asked to fit two waves per SIMD (<= 256 VGPRs; measured +6 % on Jaco2 whose general-chain state would
// otherwise take 256 + 18 parked registers); the heavier variants keep the full budget - forcing them
// under 256 spills hundreds of bytes per lane and loses 2-3x (measured).
// (use_C: only the two-pass form of orthogonal chains, abrk_ctrl.h osc_row, fits the two-wave budget; the six-row law
//  keeps its task Jacobian rows in the wavefront's LDS slab - osc_law6 - which is what lets its first pass fit)
// Runtime-table arms (the table rides in SGPRs, every frame product is a dense 3 x 4 multiply, every joint carries a
// 3 x 3 W): capped at 256 registers the three-row kernels spill 130-220 B per lane in the hot path.  One wave per SIMD
// without spills is faster (round 3, UR5's table as a user arm, 8 M rows: 908 -> 787 us; Jaco2's: 906 -> 782 us; the
// 4096-row step 8.55 -> 7.50 us).
constexpr int osc_min_waves(int km, bool use_c, int feat, bool ortho, int pass = 0, bool is_static = true) {
  if (!is_static && km <= 3) return kMinWaves;
  // six-row law: the first pass of orthogonal chains fits 256 registers (UR5: 100-216 B of scratch, all of it in cold
  // branches); general chains carry a 3 x 3 W per joint through the kinematics and would spill 250-650 B in the hot
  // path (measured in round 3: forcing them under 256 registers loses)
  // (round 6: the plain first pass of built-in / compiled GENERAL chains is asked to fit two waves as well - without the
  //  persistent loop, whose hoisted literals and parameters overflow the scalar registers (190 v_readlane per row), it needs
  //  272 + 16 registers, and capped at 256 it spills 68 B per lane: Jaco2, five task rows, same box, loop form at one wave /
  //  one row per lane at one wave / at two waves: 8 M rows 852 / 815 / 756 us, 1 M rows 133 / 127 / 107, 65 536 rows 17.3 /
  //  15.8 / 17.4, 4096-row step 9.56 / 8.44 / 9.08 us (profiles/round6/ab/jaco2_two_waves/).  With the Coriolis vector the
  //  same cap costs 370 - 430 B of scratch: those kernels keep the loop.)
  if (km == 6) return (pass == 1 && feat <= 1 && (ortho || (is_static && feat == 0 && !use_c))) ? 2 : kMinWaves;
  return ((!use_c || ortho) && feat <= 1) ? 2 : kMinWaves;
}

// Scratch of the Coriolis recursion in LDS: per link one force and one moment (6 values), laid out [link][pair][lane]
// so that every access is a 16-byte (fp64) / 8-byte (fp32) piece per lane, contiguous across the wavefront (no bank
// conflicts).  6 N values per lane: 18 KiB per wavefront for a six-joint arm in fp64 - eight wavefronts per CU (two
// per SIMD) fit the 160 KiB.  Rows never share data, so no barrier is needed: each lane reads what it wrote.

// The wavefront's copy of the sin/cos table (abrk_sincos_table.h; 2 KiB in fp64, 1 KiB in fp32): two entries per
// lane, from L2
template <class T>
__device__ __forceinline__ void load_sincos_table(T* tab, int lane) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  typedef T t2 __attribute__((ext_vector_type(2)));
  const d2* src = reinterpret_cast<const d2*>(&kSinCosTab[0][0]);
  t2* dst = reinterpret_cast<t2*>(tab);
  const d2 a = src[lane], b = src[lane + kBlock];
  dst[lane] = t2{(T)a.x, (T)a.y};
  dst[lane + kBlock] = t2{(T)b.x, (T)b.y};
  __syncthreads();
}
static_assert(kBlock == 64 && kSinCosN == 128, "two table entries per lane");
// scratch of kernels without a Coriolis recursion: nothing but the table pointer
template <class T, int N>
struct TabScratch : RegScratch<T, N> {
  static constexpr bool kHasTab = true;
  const void* sctab;
};
// The six-row OSC law (osc_law6) parks the six rows of the task Jacobian in the same slab once the Coriolis recursion is
// through with it: [6][(N + 1) / 2][kBlock] pairs, the same 18 KiB for a six-joint arm.  That takes the 36 values out
// of the register file for the whole law - what lets the six-row kernels hold two waves per SIMD.
template <int N>
constexpr int slab_pairs() { return 3 * N > 6 * ((N + 1) / 2) ? 3 * N : 6 * ((N + 1) / 2); }
template <class T, int N>
struct LdsScratch : ScratchBase {
  static constexpr bool kHasTab = true;
  using V2 = T __attribute__((ext_vector_type(2)));
  V2* slab;  // [N][3][kBlock] wrenches, then [6][(N + 1) / 2][kBlock] Jacobian rows
  int lane;
  const void* sctab;
  static constexpr int NP = (N + 1) / 2;
  template <int R>
  __device__ __forceinline__ void put_row(ic<R>, const T (&row)[N]) {
    sfor<NP>([&](auto k) ABRK_LAMBDA {
      constexpr int i0 = 2 * k(), i1 = 2 * k() + 1;
      slab[(R * NP + k()) * kBlock + lane] = V2{row[i0], i1 < N ? row[i1 < N ? i1 : i0] : T(0)};
    });
  }
  // A row is read where it is used, every time: each read goes through a pointer the optimiser cannot see through
  // (repeated reads of a row are not merged into one long-lived copy; the constant part of the address still folds
  // into the instruction's offset field) and is fenced (the scheduler does not gather the reads at the top of the law)
  // - left alone the compiler fetches all 6 N values up front and carries them through the law, i.e. spills them.
  template <int R>
  __device__ __forceinline__ void get_row(ic<R>, T (&row)[N]) const {
    __builtin_amdgcn_sched_barrier(0);
    // (an LDS-typed pointer: the reads stay ds_read_b128 - through the generic pointer they were flat loads, which take
    //  the slower flat path into LDS and tie their wait to the vector-memory counter as well)
    typedef __attribute__((address_space(3))) const V2 LV2;
    LV2* vs = (LV2*)(slab + lane);
    asm volatile("" : "+v"(vs));
    sfor<NP>([&](auto k) ABRK_LAMBDA {
      constexpr int i0 = 2 * k(), i1 = 2 * k() + 1;
      const V2 a = vs[(R * NP + k()) * kBlock];
      row[i0] = a.x;
      if constexpr (i1 < N) row[i1] = a.y;
    });
  }
  template <int K>
  __device__ __forceinline__ void put(ic<K>, const T (&fv)[3], const T (&tv)[3]) {
    slab[(K * 3 + 0) * kBlock + lane] = V2{fv[0], fv[1]};
    slab[(K * 3 + 1) * kBlock + lane] = V2{fv[2], tv[0]};
    slab[(K * 3 + 2) * kBlock + lane] = V2{tv[1], tv[2]};
  }
  // called once before the backward sweep: without it the compiler forwards the stored values to the loads, i.e. keeps
  // all 6 N of them in registers (and spills) - the very thing the slab is there to avoid
  // (the slab's address is an operand so that the slab counts as escaped)
  __device__ __forceinline__ void seal() const { asm volatile("" ::"v"(slab) : "memory"); }
  template <int K>
  __device__ __forceinline__ void get(ic<K>, T (&fv)[3], T (&tv)[3]) const {
    const V2 a = slab[(K * 3 + 0) * kBlock + lane], b = slab[(K * 3 + 1) * kBlock + lane],
             c = slab[(K * 3 + 2) * kBlock + lane];
    fv[0] = a.x;
    fv[1] = a.y;
    fv[2] = b.x;
    tv[0] = b.y;
    tv[1] = c.x;
    tv[2] = c.y;
  }
};

// (the worklist of deferred rows and the hand-over record: abrk_device.h, next to ScratchBase)
static_assert(kWlBlock == kBlock, "wl_capacity assumes one first-pass workgroup per kBlock rows");
constexpr unsigned kKm6GridCap = 4096;  // one-wave first pass of the six-row law: a persistent grid of at most this many blocks (a multiple of kWlLists)
// the first pass of the six-row law as one row per lane (no persistent grid): where it holds two waves per SIMD - a
// one-wave first pass (general chains) keeps the loop, whose next row hides the stores of the last
constexpr bool km6_first_pass_plain(int km, bool use_c, int feat, bool ortho, int pass, bool is_static) {
  return km == 6 && pass == 1 && osc_min_waves(km, use_c, feat, ortho, pass, is_static) >= 2;
}
// `mode`: 0 = every row start to finish; 1 = rows whose law needs the Jacobi eigen-decomposition (a truncating pinv that
// neither certificate excludes) only leave their index in the worklist `wl`; 2 = work that list off, densely packed
// (persistent grid: block b strides sub-list b mod kWlLists).  Lanes diverge, so in mode 0 one such row costs its whole wavefront
// the sweeps - with all six task rows that is most wavefronts (9340 executed instructions per row on UR5 against
// ~1700 without the sweeps).
// PASS (six-row kernels): 0 = the complete row program (modes 0 and 2: the sweeps are compiled in; one wave per SIMD),
// 1 = the first pass (mode 1) - no eigen-decomposition in the code at all, two waves per SIMD.
// NOTS (first pass of the plain six-row law only): the caller wants no training signal - see ScratchBase::kNoTs.
// Workgroup shape of the OSC kernels: single wavefronts with static LDS.  Rows never share data, so a workgroup is just
// a dispatch unit.  Workgroups of 2 - 4 wavefronts for the x,y,z kernels (dynamic LDS, one table per workgroup) were built
// and measured in round 4 to test whether the dispatch of single-wavefront workgroups (~1.1 per ns chip-wide) bounds the
// 131072-row shard of BASELINE config 4: it does not (UR5 + g + C, us per step with 1 / 2 / 4 wavefronts per workgroup:
// 131072 rows 10.53 / 10.22 / 10.43, 2^20 rows 58.8 / 60.6 / 58.7, 8 M rows 495.9 / 494.9 / 496.8), and the general form
// cost the config-sized step 3-4 % (profiles/round4/mw_ab.txt).  The form was removed in round 5.
template <class A, class T, int KM, bool USE_C>
constexpr bool osc_uses_slab() {
  return (USE_C && A::kOrtho) || KM == 6;
}
// EEF (the plain six-row law of arms with a two-wave first pass): the launch's ref_frame is the end effector -
// ScratchBase::kEeFrame.
template <class A, class T, int KM, bool USE_C, int FEAT, int PASS = 0, bool NOTS = false, bool EEF = false>
__global__ void __launch_bounds__(kBlock, osc_min_waves(KM, USE_C, FEAT, A::kOrtho, PASS, A::kStatic))
osc_kernel(A arm, OscP<T> P, long B, const T* __restrict__ qg, const T* __restrict__ dqg,
           const T* __restrict__ tg, const T* __restrict__ tvg, T* __restrict__ ierrg,
           const T* __restrict__ uneg, T* __restrict__ ug, T* __restrict__ tsg, int mode, int* __restrict__ wl,
           T* __restrict__ rec) {
  // the slab: scratch of the Coriolis recursion (orthogonal chains) and / or the row store of the six-row law
  constexpr bool kLds = osc_uses_slab<A, T, KM, USE_C>();
  using V2 = typename LdsScratch<T, A::N>::V2;
#if defined(ABRK_TIMELINE)
  const unsigned long long tl_entry = __builtin_amdgcn_s_memrealtime(), tl_clk0 = __builtin_amdgcn_s_memtime();
#endif
  __shared__ T sctab[2 * kSinCosN];
  const int lane = (int)threadIdx.x;
  // x,y,z kernels: the row's inputs are requested BEFORE the table fill - table and inputs come back in one memory round
  // trip instead of two in a row (profiles/round5/shard_step_timeline.md: 0.13 us + 0.2 us of a 2.5 us wavefront at 4096
  // rows, where one wavefront per SIMD hides nothing)
  OscPre<T, A::N> pre;
  const long b_xyz = (long)blockIdx.x * kBlock + threadIdx.x;
  if constexpr (KM <= 3) {
    if (b_xyz < B) osc_prefetch<A, T, KM, USE_C, FEAT>(b_xyz, qg, dqg, tg, pre);
  }
  load_sincos_table(sctab, lane);  // every lane takes part: before any exit
#if defined(ABRK_TIMELINE)
  const unsigned long long tl_tab = __builtin_amdgcn_s_memrealtime();
#endif
  __shared__ V2 slab[kLds ? slab_pairs<A::N>() * kBlock : 1];
  const bool handover = rec != nullptr;
  auto row = [&](long b, bool allow_defer) ABRK_LAMBDA {
    auto go = [&](auto& scr) ABRK_LAMBDA {
#if defined(ABRK_TIMELINE)
      if constexpr (KM <= 3) {  // stamps 0 .. 7 of this wavefront leave through `wl` (unused by the x,y,z kernels)
        osc_body<A, T, KM, USE_C, FEAT>(b, arm, P, B, qg, dqg, tg, tvg, ierrg, uneg, ug, tsg, scr, &pre);
        ABRK_STAMP(scr, 5, false);
        ABRK_STAMP(scr, 6, true);
        scr.tl[0] = tl_entry;
        scr.tl[1] = tl_tab;
        scr.tl[7] = __builtin_amdgcn_s_memtime() - tl_clk0;  // shader-clock cycles from entry to the end
        if (wl && threadIdx.x == 0) {
          unsigned long long* o = reinterpret_cast<unsigned long long*>(wl) + (size_t)blockIdx.x * 8;
          for (int k = 0; k < 8; k++) o[k] = scr.tl[k];
        }
        return false;
      }
#endif
      if constexpr (KM <= 3) {  // (inputs requested ahead of the table fill, above)
        osc_body<A, T, KM, USE_C, FEAT>(b, arm, P, B, qg, dqg, tg, tvg, ierrg, uneg, ug, tsg, scr, &pre);
        return false;
      }
      scr.allow_defer = allow_defer;
      if (allow_defer) {  // where a deferring row parks itself (ScratchBase::claim, called by the law)
        scr.wl = wl;
        scr.rec_base = rec;
        scr.handover = handover;
        scr.wl_sub = (int)(blockIdx.x % kWlLists);
        scr.wl_cap = wl_capacity(B);
        scr.row = b;
      }
      osc_body<A, T, KM, USE_C, FEAT>(b, arm, P, B, qg, dqg, tg, tvg, ierrg, uneg, ug, tsg, scr);
#if defined(ABRK_TIMELINE)
      // (timeline build, six-row first pass: stamps 0 .. 15 of this wavefront leave through the kernel's `uneg` pointer -
      //  the caller-evaluated null signal, which FEAT < 2 kernels never read)
      if constexpr (KM == 6 && PASS == 1 && FEAT < 2) {
        ABRK_STAMP(scr, 5, false);
        ABRK_STAMP(scr, 6, true);
        scr.tl[0] = tl_entry;
        scr.tl[1] = tl_tab;
        scr.tl[7] = __builtin_amdgcn_s_memtime() - tl_clk0;
        if (uneg && __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0) {
          unsigned long long* o = reinterpret_cast<unsigned long long*>(const_cast<T*>(uneg)) + (size_t)blockIdx.x * 16;
          for (int k = 0; k < 16; k++) o[k] = scr.tl[k];
        }
      }
#endif
      return scr.deferred;
    };
    static_assert(!NOTS || (KM == 6 && FEAT == 0), "NOTS is instantiated for the plain six-row law");
    static_assert(!EEF || (A::kOrtho && KM == 6 && FEAT == 0 && km6_first_pass_plain(KM, USE_C, FEAT, A::kOrtho, 1, A::kStatic)),
                  "EEF is instantiated for the plain six-row law of arms with a two-wave first pass");
    using S0 = std::conditional_t<kLds, LdsScratch<T, A::N>, TabScratch<T, A::N>>;
    using S1 = std::conditional_t<PASS == 1, DeferOnly<S0>, S0>;
    using S2 = std::conditional_t<NOTS, NoTs<S1>, S1>;
    std::conditional_t<EEF, EeFrame<S2>, S2> scr;
    if constexpr (kLds) {
      scr.slab = slab;
      scr.lane = lane;
    }
    scr.sctab = sctab;
    return go(scr);
  };
  // hand-over mode: which rows of a 64-row chunk deferred, as one mask per chunk (every chunk writes its mask, so the
  // array needs no clearing between calls); the finish kernel compacts the masks
  auto note = [&](long b, bool deferred) ABRK_LAMBDA {
    if (handover) {
      const unsigned long long m = __ballot(deferred);
      if (threadIdx.x == 0) reinterpret_cast<unsigned long long*>(wl)[b / kBlock] = m;
    }
  };
  if constexpr (km6_first_pass_plain(KM, USE_C, FEAT, A::kOrtho, PASS, A::kStatic)) {
    // first pass at two waves per SIMD: one row per lane, no loop.  The second wavefront of the SIMD hides a row's
    // memory round trips, and outside a loop nothing is hoisted: in the persistent-loop form the compiler keeps the
    // row program's literals and the controller's parameters in scalar registers across iterations, runs out of them
    // (116 spilled to vector-register lanes: 190 v_readlane per row, each a vector-ALU slot) and holds 40 hoisted
    // vector constants on top.
    long b = (long)blockIdx.x * kBlock + threadIdx.x;
    const bool in = b < B;
    const bool deferred = in && row(b, true);
    note((long)blockIdx.x * kBlock, deferred);
  } else if constexpr (KM == 6) {
    // one call site for the row program (it is inlined): modes 0 / 1 launch one lane per row and leave the loop after
    // their row, mode 2 strides a persistent grid over the worklist.  (Only the six-row kernels defer: with x,y,z
    // alone the two certificates leave < 0.01 % of the rows to the sweeps, and the loop form costs the three-row
    // kernels 40 registers.)
    // Large batches run as a persistent grid (Launch::osc_launch caps it at kKm6GridCap blocks): the kernel holds
    // one wavefront per SIMD (390-490 registers), so nothing hides a row's memory round trips but its neighbours in
    // time - in the loop a row's stores drain under the next row's arithmetic (UR5, 8 M rows: 1133 -> 985 us).
    // Requesting row i + step's inputs before working on row i was measured too: 36 more live registers cost more
    // accumulator-register traffic than the hidden latency returns (1049 us against 1004 us on one box).
    const bool list = mode == 2;
    const int sub = (int)(blockIdx.x % kWlLists);
    const int* rows = wl + 16 * kWlLists + sub * wl_capacity(B);
    const long n = list ? (long)wl[16 * sub] : B;
    const long first = list ? (long)(blockIdx.x / kWlLists) * kBlock + threadIdx.x : (long)blockIdx.x * kBlock + threadIdx.x;
    const long step = list ? (long)(gridDim.x / kWlLists) * kBlock : (long)gridDim.x * kBlock;
    for (long i = first; i < n; i += step) {
      const bool deferred = row(list ? (long)rows[i] : i, mode == 1);
      // (hand-over mode: the lanes still in the loop are a prefix of the chunk - lane 0 is among them or the chunk is
      //  past the end - so one ballot is the chunk's mask)
      if (mode == 1) note(i - threadIdx.x, deferred);
    }
  } else {
    if (b_xyz >= B) return;
    row(b_xyz, false);
  }
}

// ---- second pass of the six-row law on hand-over records (osc_law6's deferral branch wrote them; abrk_device.h rec_*,
// ScratchBase::record: the records of a 64-row chunk packed at the chunk's first slots, each carrying its row's index;
// the arithmetic: abrk_ctrl.h osc6_rec_solve).  Which rows deferred arrives as one 64-bit mask per 64-row chunk (the
// first pass's ballot).  The grid is (chunks, slots): wavefront (j, s) asks for chunk j's mask AND for the record in the
// chunk's slot s at once - one memory round trip; whether there is such a record it learns from the mask's popcount.
// (Round 4 began with a global compaction of all masks in every wavefront - a prefix sum, a second walk over the masks,
// a row list in LDS - and the dependent chain mask -> row -> record: ~1.5 us of an 8.9 us kernel.)  Two forms, chosen
// per chunk from its own count:
//   * wave-cooperative (count <= coop_rounds x slots): wavefront (j, s) takes the chunk's records s, s + slots, ...
//     Every lane decomposes the record's 6 x 6 Mx_inv - redundantly, so nothing crosses lanes and every data-dependent
//     branch of the QL iteration is uniform (only the rotations that exist are executed: ~35 of the 68 slots the
//     predicated per-lane form walks) - and applies the transformations to ITS column of [J | u_task | J v]; lanes N
//     and N + 1 then hand their column to the others (v_readlane) and lane c < N finishes joint c.  A lone lane's
//     eigen-decomposition was the critical path of every small six-row step (4096 rows: 95 % of the 64 wavefronts
//     have a truncating row, 21 us per step of which ~15 us are ONE lane's 4800 dependent instructions); here the
//     per-lane work is the scalar recurrence plus one vector, and a 4096-row step's ~190 such rows run on 190 of the
//     1024 SIMDs at once.
//   * one record per lane (more than that: arms whose Mx_inv always truncates, large batches where `slots` is small):
//     wavefront (j, 0) takes all of the chunk's records, the same arithmetic with all N + 2 columns on the lane.
// Both are the same solver with contraction pinned off: a row's bits do not depend on which ran.
template <class T>
__device__ __forceinline__ T lane_bcast(T v, int src) {
  if constexpr (sizeof(T) == 8) {
    const long long x = __builtin_bit_cast(long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(x & 0xffffffffLL), src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(x >> 32), src);
    return __builtin_bit_cast(T, (long long)(((unsigned long long)hi << 32) | lo));
  } else {
    return __builtin_bit_cast(T, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
  }
}
// (keeps a value that was asked for ahead of a branch from being asked for behind it)
template <class T>
__device__ __forceinline__ void pin_loaded(T& v) {
  asm volatile("" : "+v"(v));
}
// Workgroups are single wavefronts: the wave-cooperative form is bound by ONE wavefront's instruction stream, and two
// of them on a SIMD halve each other's issue rate (measured with 512-thread workgroups: eight working wavefronts per
// CU, 14.4 us for 190 rows; the dispatcher spreads single-wavefront workgroups over the CUs).  blockIdx.x is the chunk:
// the wavefronts that have a record (slots 0, 1, 2 of most chunks) come first in dispatch order and spread over all XCDs.
template <int N, class T>
__global__ void __launch_bounds__(kBlock)
osc6_finish_kernel(const unsigned long long* __restrict__ masks, const T* __restrict__ recs, int nulls, int coop_rounds,
                   long B, T* __restrict__ ug, T* __restrict__ tsg) {
  const int lane = (int)threadIdx.x;
  const long j = blockIdx.x;
  const int s0 = (int)blockIdx.y, slots = (int)gridDim.y;
  const int c = lane < N + 2 ? lane : N + 1;  // (idle lanes shadow the last column)
  const int jc = lane < N ? lane : 0;
  const T* rec = recs + (j * kBlock + s0) * rec_len(N);
  // mask and record together (the record's slot exists whatever it holds: the host sizes `recs` in whole chunks)
  // (the mask through the vector memory path, like the record's columns: as a scalar load the compiler queues it behind
  //  the wait for the record's scalar loads - two round trips again)
  long jv = j;
  pin_loaded(jv);
  unsigned long long mask = masks[jv];
  T S[21], G[1][6], ridx, b1, b2;
  auto load = [&]() ABRK_LAMBDA {
    osc6_rec_load<N, T, 1>(rec, c, S, G);
    ridx = rec[21];
    // the two joint-space sums are asked for with the rest of the record: one memory round trip, not two
    b1 = rec[rec_off_b1(N) + jc];
    b2 = rec[rec_off_b1(N) + N + jc];
  };
  load();
  sfor<21>([&](auto e) ABRK_LAMBDA { pin_loaded(S[e()]); });
  sfor<6>([&](auto r) ABRK_LAMBDA { pin_loaded(G[0][r()]); });
  pin_loaded(ridx);
  pin_loaded(b1);
  pin_loaded(b2);
  pin_loaded(mask);
  const int cnt = __builtin_amdgcn_readfirstlane(__popcll(mask));
  if (s0 >= cnt) return;  // nothing in this slot
  if (cnt <= coop_rounds * slots) {
    for (int s = s0;;) {
      // (a record's row index is data: whatever a slot holds, nothing is stored outside [0, B))
      const bool row_ok = ridx >= T(0) && ridx < T(B);
      const long b = row_ok ? (long)ridx : 0;
      {
#pragma clang fp contract(off)  // the same bits as osc6_finish_row
        T li[6], iq[6], y[6];
        osc6_rec_solve<N, T, 1, true>(rec, c, S, G, li, iq);
        ql_pinv_solve<6>(li, iq, G[0], y);  // this lane's column through the pseudo-inverse
        T a1 = T(-0.0), a2 = T(-0.0);
        sfor<6>([&](auto i) ABRK_LAMBDA {
          const T yu = lane_bcast(y[i()], N), yw = lane_bcast(y[i()], N + 1);
          a1 = Rm<T>::fma(G[0][i()], yu, a1);
          a2 = Rm<T>::fma(G[0][i()], yw, a2);
        });
        if (lane < N && row_ok) {
          const T ts = b1 - a1;
          ug[b * N + lane] = ts + b2 - (nulls ? a2 : T(0));
          if (tsg) tsg[b * N + lane] = ts;
        }
      }
      s += slots;
      if (s >= cnt) break;
      rec = recs + (j * kBlock + s) * rec_len(N);
      load();
    }
  } else if (s0 == 0 && lane < cnt) {
    rec = recs + (j * kBlock + lane) * rec_len(N);
    const T rix = rec[21];
    if (rix >= T(0) && rix < T(B)) {
      const long b = (long)rix;
      T u[N], ts[N];
      osc6_finish_row<N, T>(rec, nulls != 0, u, ts);
      store_row<N>(ug, b, u);
      if (tsg) store_row<N>(tsg, b, ts);
    }
  }
}
// Wavefronts per chunk and records a wavefront takes at most, by batch size (random UR5 states with all six task rows:
// 4.6 % defer, 2.9 per chunk, more than 12 never).  Measured, us per step new / round-4 global compaction: 4096 rows
// 15.6 / 16.4, 8192 15.7, 16 k 19.4 / 17.4, 32 k 20.2 / 23.0, 64 k 27.6 / 36.8.  Up to 8192 rows every (chunk, slot)
// has a SIMD of its own; at 16 k rows the wavefronts of slots >= 4 share a SIMD with a working one (the finish kernel
// lasts 12.0 us instead of 8.2 - the one size where compacting over the whole batch was better); beyond that the
// working wavefronts outnumber the SIMDs anyway, a second round costs more than a second wavefront on the SIMD, and
// chunks with more records than slots go one record per lane.
inline int finish_slots(long B) {
  const long nchunk = (B + kBlock - 1) / kBlock;
  return nchunk <= 256 ? 12 : nchunk <= 512 ? 8 : 2;
}
inline int finish_rounds(long B) { return (B + kBlock - 1) / kBlock <= 256 ? 2 : 1; }
constexpr long kHandoverMaxRows = 262144;  // (the finish kernel's grid: 4096 chunks x slots)

// Mode F: u + Tx, J, M, g in one launch (840 B per UR5 row in fp64: HBM-bound).  One LDS slab serves both the
// cooperative stores and (use_C) the scratch of the Coriolis recursion, which is dead by the time the first row is
// parked.  FEAT is 0 (the plain law) or 2 (every optional input).
// (the C / dJ variant is not capped at 256 registers: that costs 364-424 B of scratch, measured in round 3)
template <class A, class T, int KM, bool USE_C, int FEAT, bool VEL = false>
__global__ void __launch_bounds__(kBlock, VEL ? kMinWaves : osc_min_waves(KM, USE_C, FEAT, A::kOrtho, 0, A::kStatic))
osc_full_kernel(A arm, OscP<T> P, long B, const T* __restrict__ qg, const T* __restrict__ dqg,
                const T* __restrict__ tg, const T* __restrict__ tvg, T* __restrict__ ierrg,
                const T* __restrict__ uneg, T* __restrict__ ug, T* __restrict__ tsg, unsigned want, DynOutP<T> out) {
  // the slab: cooperative stores, scratch of the Coriolis recursion, row store of the six-row law - one after the other
  constexpr int kSlabT = max_row(A::N) > 2 * slab_pairs<A::N>() ? max_row(A::N) : 2 * slab_pairs<A::N>();
  __shared__ __attribute__((aligned(16))) T slab[kBlock * kSlabT];
  const long row0 = (long)blockIdx.x * kBlock;
  const long b = row0 + threadIdx.x;
  LdsStore<T> st{slab, row0, B, (int)threadIdx.x};
  __shared__ T sctab[2 * kSinCosN];
  load_sincos_table(sctab, (int)threadIdx.x);
  if constexpr ((USE_C && !VEL && A::kOrtho) || KM == 6) {
    using V2 = typename LdsScratch<T, A::N>::V2;
    LdsScratch<T, A::N> scr;
    scr.slab = reinterpret_cast<V2*>(slab);
    scr.lane = (int)threadIdx.x;
    scr.sctab = sctab;
    osc_full_body<A, T, KM, USE_C, FEAT, VEL>(b, b < B, st, arm, P, B, qg, dqg, tg, tvg, ierrg, uneg, ug, tsg, want, out, scr);
  } else {
    TabScratch<T, A::N> scr;
    scr.sctab = sctab;
    osc_full_body<A, T, KM, USE_C, FEAT, VEL>(b, b < B, st, arm, P, B, qg, dqg, tg, tvg, ierrg, uneg, ug, tsg, want, out, scr);
  }
}

template <class A, class T>
__global__ void __launch_bounds__(kBlock, kMinWaves)
sliding_kernel(A arm, SlidingP<T> P, long B, const T* __restrict__ qg, const T* __restrict__ dqg,
               const T* __restrict__ tg, const T* __restrict__ tvg, const T* __restrict__ tag,
               T* __restrict__ ug, T* __restrict__ sg) {
  constexpr bool kTab = true;
  __shared__ T sctab[2 * kSinCosN];
  load_sincos_table(sctab, (int)threadIdx.x);
  // grid-stride over the row blocks (the launcher caps the grid at kSlidingMaxBlocks): a threejoint fp32 wavefront
  // lives ~5 us, and one-wavefront workgroups are dispatched at ~1.1 per ns chip-wide - too slowly to keep the 7
  // wavefronts per SIMD this kernel could hold resident (8 M rows: 131 k dispatches)
  for (long b = (long)blockIdx.x * kBlock + threadIdx.x; b < B; b += (long)gridDim.x * kBlock)
    sliding_body<A, T, kTab>(b, arm, P, B, qg, dqg, tg, tvg, tag, ug, sg, sctab);
}
constexpr long kSlidingMaxBlocks = 256L * 32 * 4;  // four rounds of a full chip of wavefronts

template <class A, class T>
__global__ void __launch_bounds__(kBlock, kMinWaves)
joint_kernel(A arm, JointP<T> P, long B, const T* __restrict__ qg, const T* __restrict__ dqg,
             const T* __restrict__ tg, const T* __restrict__ tvg, T* __restrict__ ug) {
  ABRK_ROW_INDEX
  joint_body<A, T>(b, arm, P, B, qg, dqg, tg, tvg, ug);
}

template <int N, class T>
__global__ void __launch_bounds__(kBlock, kMinWaves)
osc_law_kernel(OscP<T> P, long B, const T* __restrict__ Jg, const T* __restrict__ Mg, const T* __restrict__ gg,
               const T* __restrict__ cg, const T* __restrict__ xg, const T* __restrict__ Rg,
               const T* __restrict__ qg, const T* __restrict__ dqg, const T* __restrict__ tg,
               const T* __restrict__ tvg, T* __restrict__ ierrg, const T* __restrict__ uneg, T* __restrict__ ug,
               T* __restrict__ tsg) {
  ABRK_ROW_INDEX
  osc_law_body<N, T>(b, P, B, Jg, Mg, gg, cg, xg, Rg, qg, dqg, tg, tvg, ierrg, uneg, ug, tsg);
}

// ---------------------------------------------------------------- launch table
// Type-erased launchers the host ABI (abrk_host.cpp) calls; pointers are device pointers.
struct LaunchArgs {
  const void* arm_rt;   // RtArm<N,T> table for user arms (matching dtype), else nullptr
  long B;
  hipStream_t stream;
};
struct FinishArgs {
  const void* masks;  // one 64-bit mask per 64-row chunk: the rows the first pass deferred
  const void* rec;
  int nulls;
  int slots;        // wavefronts per 64-row chunk (finish_slots)
  int coop_rounds;  // records a wavefront takes at most; a chunk with more than slots x coop_rounds goes one record per lane
  void *u, *ts;
  int group = 0;    // > 0: the grouped form - `group` chunks share 4 x group wavefronts (abrk_law.hip osc6_finish_group_kernel)
};
// (abrk_law.hip; arm-independent: the record holds everything)
hipError_t launch_osc6_finish(int n_joints, int dtype, const LaunchArgs& la, const FinishArgs& a);
template <class A, class T, bool USE_C, int KM>
__global__ void __launch_bounds__(kBlock, kMinWaves)
rollout_kernel(A arm, OscP<T> P, TwoLinkP<T> K, long B, int n_steps, int every, T* __restrict__ qg,
               T* __restrict__ dqg, const T* __restrict__ tg, T* __restrict__ ierrg, T* __restrict__ qt,
               T* __restrict__ dqt, T* __restrict__ ut) {
  ABRK_ROW_INDEX
  rollout_body<A, T, USE_C, KM>(b, arm, P, K, B, n_steps, every, qg, dqg, tg, ierrg, qt, dqt, ut);
}

template <class T>
__global__ void __launch_bounds__(kBlock)
twolink_step_kernel(TwoLinkP<T> K, long B, T* __restrict__ qg, T* __restrict__ dqg, const T* __restrict__ ug) {
  ABRK_ROW_INDEX
  twolink_step_body<T>(b, K, qg, dqg, ug);
}

template <class A, class T>
__global__ void __launch_bounds__(kBlock, kMinWaves)
ik_kernel(A arm, IkP<T> P, long B, const T* __restrict__ qg, const T* __restrict__ tg, T* __restrict__ pp,
          T* __restrict__ vp) {
  ABRK_ROW_INDEX
  ik_body<A, T>(b, arm, P, B, qg, tg, pp, vp);
}

template <int N, class T>
__global__ void __launch_bounds__(kBlock)
limits_kernel(LimitsP<T> P, long B, const T* __restrict__ qg, T* __restrict__ ug, int acc) {
  ABRK_ROW_INDEX
  limits_body<N, T>(b, P, qg, ug, acc);
}

template <class A, class T>
__global__ void __launch_bounds__(kBlock, kMinWaves)
floating_kernel(A arm, int dynamic, int task_space, long B, const T* __restrict__ qg, const T* __restrict__ dqg,
                T* __restrict__ ug, int acc) {
  ABRK_ROW_INDEX
  floating_body<A, T>(b, arm, dynamic, task_space, qg, dqg, ug, acc);
}

// (capped at 256 VGPRs for two waves per SIMD - UR5 fp64: 320 -> 256 + 172 B of scratch - the one-pass kernel measured no
// faster at 8 M rows, 2830 vs 2822 us: it is issue-bound, and slower at 4096, 38.5 vs 30.2 us)
template <class A, class T>
__global__ void __launch_bounds__(kBlock, kMinWaves)
obstacles_kernel(A arm, ObsP<T> P, long B, const T* __restrict__ qg, T* __restrict__ ug, int acc) {
  // grid-stride (the launcher caps the grid at kObstaclesMaxBlocks): one wavefront per SIMD in fp64, so in the loop a
  // row's stores drain under the next row's arithmetic, as in the six-row OSC kernels
  for (long b = (long)blockIdx.x * kBlock + threadIdx.x; b < B; b += (long)gridDim.x * kBlock)
    obstacles_body<A, T>(b, arm, P, qg, ug, acc);
}
constexpr long kObstaclesMaxBlocks = 4096;
// AvoidObstacles with the heavy (obstacle, segment) pairs of a wavefront's 64 rows redistributed over its lanes through
// LDS (abrk_ctrl.h obstacles_phase_a / obstacles_pair): orthogonal chains of three joints and more, up to 64 heavy slots
// (obstacles x (N - 2) segments).  LDS per wavefront: the rows' records [field][lane] (66 values per row for six joints:
// 33 KiB in fp64), the pair list and one round of contributions - 40 KiB, four wavefronts per CU.
constexpr int kObsPairCap = 2048;  // pairs of a wavefront that are redistributed; a lane beyond works its own off
template <class A, class T>
constexpr bool obstacles_use_lds() { return A::kOrtho && A::N >= 3; }
template <class A, class T>
__global__ void __launch_bounds__(kBlock, kMinWaves)
obstacles_lds_kernel(A arm, ObsP<T> P, long B, const T* __restrict__ qg, T* __restrict__ ug, int acc) {
  constexpr int N = A::N, RL = obs_rec_len<N>();
  __shared__ T rec[RL * kBlock];
  __shared__ unsigned short pairs[kObsPairCap];
  __shared__ T contrib[kBlock * N];
  const int lane = (int)threadIdx.x;
  for (long b0 = (long)blockIdx.x * kBlock; b0 < B; b0 += (long)gridDim.x * kBlock) {
    const long b = b0 + lane;
    const bool in = b < B;
    T u[N];
    unsigned long long heavy = 0ull;
    if (in) {
      T q[N];
      load_row<N>(qg, b, q);
      heavy = obstacles_phase_a<A, T>(arm, P, q, u, [&](int f, T v) ABRK_LAMBDA { rec[f * kBlock + lane] = v; });
    }
    // number the wavefront's pairs: lane l owns [pre, pre + cnt), in slot order
    const int cnt = __popcll(heavy);
    int incl = cnt;
    for (int d = 1; d < kBlock; d <<= 1) {
      const int v = __shfl_up(incl, d);
      if (lane >= d) incl += v;
    }
    const int total = __builtin_amdgcn_readlane(incl, kBlock - 1);
    const int pre = incl - cnt;
    const int shared = total < kObsPairCap ? total : kObsPairCap;  // pairs [0, shared) are redistributed
    unsigned long long overflow = 0ull;
    {
      int p = pre;
      unsigned long long bits = heavy;
      while (bits && p < kObsPairCap) {
        const int slot = __builtin_ctzll(bits);
        bits &= bits - 1;
        pairs[p++] = (unsigned short)(lane | (slot << 6));
      }
      overflow = bits;
    }
    __syncthreads();
    for (int r0 = 0; r0 < shared; r0 += kBlock) {
      const int p = r0 + lane;
      if (p < shared) {
        const int e = pairs[p], src = e & (kBlock - 1), slot = e >> 6;
        T c[N];
        obstacles_pair<N, T>(P, slot, [&](int f) ABRK_LAMBDA { return rec[f * kBlock + src]; }, c);
        sfor<N>([&](auto i) ABRK_LAMBDA { contrib[lane * N + i()] = c[i()]; });
      }
      __syncthreads();
      // every owner adds its pairs of this round, in slot order
      const int lo = pre > r0 ? pre : r0, hi = (pre + cnt < r0 + kBlock ? pre + cnt : r0 + kBlock) < shared ? (pre + cnt < r0 + kBlock ? pre + cnt : r0 + kBlock) : shared;
      for (int pp = lo; pp < hi; pp++) sfor<N>([&](auto i) ABRK_LAMBDA { u[i()] += contrib[(pp - r0) * N + i()]; });
      __syncthreads();
    }
    // what did not fit the list (a wavefront with more than 32 heavy pairs per row on average) stays with its row - its
    // HIGH slots, added after the redistributed low ones: the slot order of the one-pass kernel (the records in LDS are
    // untouched until the next block of rows)
    while (overflow) {
      const int slot = __builtin_ctzll(overflow);
      overflow &= overflow - 1;
      T c[N];
      obstacles_pair<N, T>(P, slot, [&](int f) ABRK_LAMBDA { return rec[f * kBlock + lane]; }, c);
      sfor<N>([&](auto i) ABRK_LAMBDA { u[i()] += c[i()]; });
    }
    if (in) {
      obstacles_finish<N, T>(P, u);
      put_row<N>(ug, b, u, acc);
    }
  }
}

template <int N, class T>
__global__ void __launch_bounds__(kBlock)
mx_kernel(long B, int k, T thr, const T* __restrict__ Mg, const T* __restrict__ Jg, T* __restrict__ Mxg,
          T* __restrict__ Minvg) {
  ABRK_ROW_INDEX
  mx_body<N, T>(b, k, thr, Mg, Jg, Mxg, Minvg);
}
template <class T>
__global__ void __launch_bounds__(kBlock)
velocity_limiting_kernel(long B, T kp, T ko, T kv, T vmax0, T vmax1, const T* __restrict__ ing, T* __restrict__ outg) {
  ABRK_ROW_INDEX
  velocity_limiting_body<T>(b, kp, ko, kv, vmax0, vmax1, ing, outg);
}
template <class T>
__global__ void __launch_bounds__(kBlock)
orientation_forces_kernel(long B, int alg, const T* __restrict__ Rg, const T* __restrict__ ag, T* __restrict__ outg) {
  ABRK_ROW_INDEX
  orientation_forces_body<T>(b, alg, Rg, ag, outg);
}
template <class T>
__global__ void __launch_bounds__(kBlock)
transformations_kernel(long B, int op, const T* __restrict__ ag, const T* __restrict__ bg, T* __restrict__ outg) {
  ABRK_ROW_INDEX
  transformations_body<T>(b, op, ag, bg, outg);
}
hipError_t launch_transformations(int dtype, const LaunchArgs& la, int op, const void* a, const void* b, void* out);
// arm-independent helpers of the OSC law (abrk_law.hip)
hipError_t launch_osc_mx(int n_joints, int dtype, const LaunchArgs& la, int k, double threshold, const void* M,
                         const void* J, void* Mx, void* Minv);
hipError_t launch_velocity_limiting(int dtype, const LaunchArgs& la, const double (&gains)[5], const void* in,
                                    void* out);
hipError_t launch_orientation_forces(int dtype, const LaunchArgs& la, int alg, const void* R, const void* abg,
                                     void* out);

struct FloatingArgs {
  int dynamic, task_space, acc;
  const void *q, *dq;
  void* u;
};
struct ObstaclesArgs {
  const void* P;  // ObsP<T>
  int acc;
  const void* q;
  void* u;
};
// AvoidJointLimits needs no arm, only the joint count (abrk_law.hip)
hipError_t launch_limits(int n_joints, int dtype, const LaunchArgs& la, const void* P, const void* q, void* u,
                         int acc);

struct IkArgs {
  const void* P;  // IkP<T>
  const void *q, *target;
  void *pp, *vp;
};

struct RolloutArgs {
  const void *P, *K;  // OscP<T>, TwoLinkP<T>
  int use_C, fast, n_steps, every;
  void *q, *dq;
  const void* target;
  void *ierr, *qt, *dqt, *ut;
};

struct LawArgs {
  const void* P;  // OscP<T>
  const void *J, *M, *g, *c, *xyz, *R, *q, *dq, *target, *tv, *une;
  void *ierr, *u, *ts;
};
hipError_t launch_osc_law(int n_joints, int dtype, const LaunchArgs& la, const LawArgs& a);  // abrk_law.hip

struct DynArgs {
  int frame, m;
  double off[3];
  unsigned want;
  const void *q, *dq;
  void* out[10];
};
struct OscArgs {
  const void* P;  // OscP<T>
  int fast, use_C;  // fast: 0 general six-row kernel, 2 / 3: first two / three position rows of the EE
  const void *q, *dq, *target, *tv, *une;
  void *ierr, *u, *ts;
  int* wl = nullptr;  // worklist of B + 1 ints: rows that need the Jacobi sweeps are deferred to a dense second pass
  // hand-over records (rec_len(N) values per ROW): the first pass alone is launched, a deferred row leaves its record,
  // `wl` receives one 64-bit mask per 64-row chunk, and the CALLER enqueues launch_osc6_finish behind it
  void* rec = nullptr;
  unsigned want = 0;  // != 0: the fused Mode-F kernel also writes Tx / J / M / g / C / dJ (W_TX | ... | W_DJ)
  void* out[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};
struct SlidingArgs {
  const void* P;  // SlidingP<T>
  const void *q, *dq, *target, *tv, *ta;
  void *u, *s;
};
struct JointArgs {
  const void* P;  // JointP<T>
  const void *q, *dq, *target, *tv;
  void* u;
};
struct ArmOps {
  int n_joints;
  hipError_t (*dyn)(int dtype, const LaunchArgs&, const DynArgs&);
  hipError_t (*osc)(int dtype, const LaunchArgs&, const OscArgs&);
  hipError_t (*sliding)(int dtype, const LaunchArgs&, const SlidingArgs&);
  hipError_t (*joint)(int dtype, const LaunchArgs&, const JointArgs&);
  hipError_t (*rollout)(int dtype, const LaunchArgs&, const RolloutArgs&);  // two-joint arms only, else null
  hipError_t (*ik)(int dtype, const LaunchArgs&, const IkArgs&);
  hipError_t (*floating)(int dtype, const LaunchArgs&, const FloatingArgs&);
  hipError_t (*obstacles)(int dtype, const LaunchArgs&, const ObstaclesArgs&);
};
hipError_t launch_twolink_step(int dtype, const LaunchArgs& la, const void* K, void* q, void* dq, const void* u);

inline dim3 grid_for(long B) { return dim3((unsigned)((B + kBlock - 1) / kBlock)); }
template <class A, class T>
struct Launch {
  static A arm_of(const LaunchArgs& la) {
    if constexpr (A::kStatic) return A{};
    else return *static_cast<const A*>(la.arm_rt);
  }
  static hipError_t dyn(const LaunchArgs& la, const DynArgs& a) {
    DynOutP<T> o;
    T** po = reinterpret_cast<T**>(&o);
    for (int i = 0; i < 10; i++) po[i] = static_cast<T*>(a.out[i]);
    A arm = arm_of(la);
    if (a.want & (W_C | W_DJ))
      hipLaunchKernelGGL((dyn_kernel<A, T, true>), grid_for(la.B), dim3(kBlock), 0, la.stream, arm, a.frame, a.m,
                         T(a.off[0]), T(a.off[1]), T(a.off[2]), a.want, la.B, (const T*)a.q, (const T*)a.dq, o);
    else
      hipLaunchKernelGGL((dyn_kernel<A, T, false>), grid_for(la.B), dim3(kBlock), 0, la.stream, arm, a.frame, a.m,
                         T(a.off[0]), T(a.off[1]), T(a.off[2]), a.want, la.B, (const T*)a.q, (const T*)a.dq, o);
    return hipGetLastError();
  }
  template <int KM, bool UC, int FEAT>
  static hipError_t osc_launch(const LaunchArgs& la, const OscArgs& a) {
    // the first pass of the plain six-row law has an instantiation for "ref_frame is the end effector" (EEF: no frame
    // capture in the forward kinematics; the same bits) - the reference benchmark's setting
    // Built for orthogonal built-in / compiled chains (where it was measured, UR5 8 M rows -1.6 %); general chains keep the
    // capture (their EEF first pass faulted on the four-joint test arm: profiles/round6/NOTES.md section 2).  EVERY
    // pass of such a launch takes the EEF form (first pass, recompute pass, one-pass): the capture changes the basic-block
    // structure of the forward kinematics, and with it which multiply the compiler fuses with which add in its
    // `c a0 + s a1` shapes - a row's bits must not depend on the pass that evaluates it.
    constexpr bool kEefBuilt = A::kOrtho && km6_first_pass_plain(KM, UC, FEAT, A::kOrtho, 1, A::kStatic);
    const bool eef = kEefBuilt && static_cast<const OscP<T>*>(a.P)->ref_frame == 2 * A::N + 1;
    auto launch = [&](auto pass, auto nots_, auto eef_, dim3 grid, int mode) {
      hipLaunchKernelGGL((osc_kernel<A, T, KM, UC, FEAT, pass(), nots_(), eef_()>), grid, dim3(kBlock), 0, la.stream,
                         arm_of(la), *static_cast<const OscP<T>*>(a.P), la.B, (const T*)a.q, (const T*)a.dq,
                         (const T*)a.target, (const T*)a.tv, (T*)a.ierr, (const T*)a.une, (T*)a.u,
                         nots_() ? (T*)nullptr : (T*)a.ts, mode, a.wl, (T*)a.rec);
    };
    using std::false_type;
    using std::true_type;
    auto go = [&](auto pass, dim3 grid, int mode) {
      if constexpr (kEefBuilt && FEAT == 0) {
        if (eef) return launch(pass, false_type{}, true_type{}, grid, mode);
      }
      launch(pass, false_type{}, false_type{}, grid, mode);
    };
    // the plain six-row law when no training signal is asked for (NOTS: gravity joins the velocity term before the
    // factorisations).  EVERY form of the law then runs that arithmetic - first pass, recompute pass, one-pass - so
    // that a row's bits do not depend on the batch it arrives in.
    auto go_nots = [&](auto pass, dim3 grid, int mode) {
      if constexpr (KM == 6 && FEAT == 0) {
        if constexpr (kEefBuilt) {
          if (eef) return launch(pass, true_type{}, true_type{}, grid, mode);
        }
        launch(pass, true_type{}, false_type{}, grid, mode);
      }
    };
    const bool nots = KM == 6 && FEAT == 0 && !a.ts;
    if constexpr (KM == 6) {
      if (a.wl) {
        // stale counters would let pass 1 append past its sub-lists: no launch without the memset (hand-over mode: no
        // counters - every chunk writes its mask)
        if (!a.rec)
          if (hipError_t e = hipMemsetAsync(a.wl, 0, 16 * kWlLists * sizeof(int), la.stream); e != hipSuccess) return e;
        dim3 g1 = grid_for(la.B);
        // (the persistent-grid form of the first pass: a multiple of kWlLists)
        constexpr bool plain = km6_first_pass_plain(KM, UC, FEAT, A::kOrtho, 1, A::kStatic);
        if (!plain && g1.x > kKm6GridCap) g1.x = kKm6GridCap;
        if (nots) go_nots(ic<1>{}, g1, 1);
        else go(ic<1>{}, g1, 1);
        if (!a.rec) {  // a multiple of kWlLists: 8 blocks stride each sub-list
          if (nots) go_nots(ic<0>{}, dim3(8 * kWlLists), 2);
          else go(ic<0>{}, dim3(8 * kWlLists), 2);
        }
        return hipSuccess;
      }
    }
    if (nots) go_nots(ic<0>{}, grid_for(la.B), 0);
    else go(ic<0>{}, grid_for(la.B), 0);
    return hipSuccess;
  }
  template <int KM, bool UC>
  static hipError_t osc_launch_feat(const LaunchArgs& la, const OscArgs& a) {
    // which optional inputs are present?  0: none, 1: fused null controllers only, 2: anything else
    const bool other = a.tv || a.ierr || a.une;
    const bool nulls = static_cast<const OscP<T>*>(a.P)->n_null > 0;
    if (other) return osc_launch<KM, UC, 2>(la, a);
    if (nulls) return osc_launch<KM, UC, 1>(la, a);
    return osc_launch<KM, UC, 0>(la, a);
  }
  template <int KM, bool UC, int FEAT, bool VEL>
  static void osc_full_launch(const LaunchArgs& la, const OscArgs& a) {
    DynOutP<T> o{};
    o.Tx = (T*)a.out[0];
    o.J = (T*)a.out[1];
    o.M = (T*)a.out[2];
    o.g = (T*)a.out[3];
    o.C = (T*)a.out[4];
    o.dJ = (T*)a.out[5];
    hipLaunchKernelGGL((osc_full_kernel<A, T, KM, UC, FEAT, VEL>), grid_for(la.B), dim3(kBlock), 0, la.stream, arm_of(la),
                       *static_cast<const OscP<T>*>(a.P), la.B, (const T*)a.q, (const T*)a.dq, (const T*)a.target,
                       (const T*)a.tv, (T*)a.ierr, (const T*)a.une, (T*)a.u, (T*)a.ts, a.want, o);
  }
  template <int KM, bool UC>
  static void osc_full_feat(const LaunchArgs& la, const OscArgs& a) {
    const bool plain = !(a.tv || a.ierr || a.une) && static_cast<const OscP<T>*>(a.P)->n_null == 0;
    // C / dJ among the outputs: the variant whose dynamics pass assembles the Christoffel matrix (FEAT 2 only: the
    // velocity-dependent outputs are the rarer request and one instantiation per (KM, use_C) keeps the build in bounds)
    if (a.want & (W_C | W_DJ)) osc_full_launch<KM, UC, 2, true>(la, a);
    else if (plain) osc_full_launch<KM, UC, 0, false>(la, a);
    else osc_full_launch<KM, UC, 2, false>(la, a);
  }
  static hipError_t osc_full(const LaunchArgs& la, const OscArgs& a) {
    // the two-row kernel of the planar examples is not duplicated: x,y control of a small arm takes the six-row form
    if (a.fast == 3) {
      if (a.use_C) osc_full_feat<3, true>(la, a);
      else osc_full_feat<3, false>(la, a);
    } else {
      if (a.use_C) osc_full_feat<6, true>(la, a);
      else osc_full_feat<6, false>(la, a);
    }
    return hipGetLastError();
  }
  static hipError_t osc(const LaunchArgs& la, const OscArgs& a) {
    if (a.want) return osc_full(la, a);
    hipError_t e = hipSuccess;
    if (a.fast == 3) {
      e = a.use_C ? osc_launch_feat<3, true>(la, a) : osc_launch_feat<3, false>(la, a);
    } else if (a.fast == 2 && A::N <= 3) {
      if constexpr (A::N <= 3) e = a.use_C ? osc_launch_feat<2, true>(la, a) : osc_launch_feat<2, false>(la, a);
    } else {
      e = a.use_C ? osc_launch_feat<6, true>(la, a) : osc_launch_feat<6, false>(la, a);
    }
    return e != hipSuccess ? e : hipGetLastError();
  }
  static hipError_t sliding(const LaunchArgs& la, const SlidingArgs& a) {
    const long blocks = (la.B + kBlock - 1) / kBlock;
    hipLaunchKernelGGL((sliding_kernel<A, T>), dim3((unsigned)(blocks < kSlidingMaxBlocks ? blocks : kSlidingMaxBlocks)),
                       dim3(kBlock), 0, la.stream, arm_of(la),
                       *static_cast<const SlidingP<T>*>(a.P), la.B, (const T*)a.q, (const T*)a.dq,
                       (const T*)a.target, (const T*)a.tv, (const T*)a.ta, (T*)a.u, (T*)a.s);
    return hipGetLastError();
  }
  static hipError_t joint(const LaunchArgs& la, const JointArgs& a) {
    hipLaunchKernelGGL((joint_kernel<A, T>), grid_for(la.B), dim3(kBlock), 0, la.stream, arm_of(la),
                       *static_cast<const JointP<T>*>(a.P), la.B, (const T*)a.q, (const T*)a.dq, (const T*)a.target,
                       (const T*)a.tv, (T*)a.u);
    return hipGetLastError();
  }
};

// ops for an arm policy available in both arithmetic types (AD = double flavour, AF = float)
template <class AD, class AF>
struct OpsFor {
  static hipError_t dyn(int dt, const LaunchArgs& la, const DynArgs& a) {
    return dt == 0 ? Launch<AD, double>::dyn(la, a) : Launch<AF, float>::dyn(la, a);
  }
  static hipError_t osc(int dt, const LaunchArgs& la, const OscArgs& a) {
    return dt == 0 ? Launch<AD, double>::osc(la, a) : Launch<AF, float>::osc(la, a);
  }
  static hipError_t sliding(int dt, const LaunchArgs& la, const SlidingArgs& a) {
    return dt == 0 ? Launch<AD, double>::sliding(la, a) : Launch<AF, float>::sliding(la, a);
  }
  static hipError_t joint(int dt, const LaunchArgs& la, const JointArgs& a) {
    return dt == 0 ? Launch<AD, double>::joint(la, a) : Launch<AF, float>::joint(la, a);
  }
  template <class A, class T>
  static hipError_t rollout_t(const LaunchArgs& la, const RolloutArgs& a) {
    if constexpr (A::N == 2) {
      A arm = Launch<A, T>::arm_of(la);
      auto go = [&](auto uc, auto km) {
        hipLaunchKernelGGL((rollout_kernel<A, T, uc(), km()>), grid_for(la.B), dim3(kBlock), 0, la.stream, arm,
                           *static_cast<const OscP<T>*>(a.P), *static_cast<const TwoLinkP<T>*>(a.K), la.B, a.n_steps,
                           a.every, (T*)a.q, (T*)a.dq, (const T*)a.target, (T*)a.ierr, (T*)a.qt, (T*)a.dqt, (T*)a.ut);
      };
      using std::integral_constant;
      if (a.fast == 2) {
        if (a.use_C) go(integral_constant<bool, true>{}, integral_constant<int, 2>{});
        else go(integral_constant<bool, false>{}, integral_constant<int, 2>{});
      } else {
        if (a.use_C) go(integral_constant<bool, true>{}, integral_constant<int, 6>{});
        else go(integral_constant<bool, false>{}, integral_constant<int, 6>{});
      }
      return hipGetLastError();
    } else {
      return hipErrorInvalidValue;
    }
  }
  static hipError_t rollout(int dt, const LaunchArgs& la, const RolloutArgs& a) {
    return dt == 0 ? rollout_t<AD, double>(la, a) : rollout_t<AF, float>(la, a);
  }
  template <class A, class T>
  static hipError_t ik_t(const LaunchArgs& la, const IkArgs& a) {
    hipLaunchKernelGGL((ik_kernel<A, T>), grid_for(la.B), dim3(kBlock), 0, la.stream, Launch<A, T>::arm_of(la),
                       *static_cast<const IkP<T>*>(a.P), la.B, (const T*)a.q, (const T*)a.target, (T*)a.pp, (T*)a.vp);
    return hipGetLastError();
  }
  static hipError_t ik(int dt, const LaunchArgs& la, const IkArgs& a) {
    return dt == 0 ? ik_t<AD, double>(la, a) : ik_t<AF, float>(la, a);
  }
  template <class A, class T>
  static hipError_t floating_t(const LaunchArgs& la, const FloatingArgs& a) {
    hipLaunchKernelGGL((floating_kernel<A, T>), grid_for(la.B), dim3(kBlock), 0, la.stream, Launch<A, T>::arm_of(la),
                       a.dynamic, a.task_space, la.B, (const T*)a.q, (const T*)a.dq, (T*)a.u, a.acc);
    return hipGetLastError();
  }
  static hipError_t floating(int dt, const LaunchArgs& la, const FloatingArgs& a) {
    return dt == 0 ? floating_t<AD, double>(la, a) : floating_t<AF, float>(la, a);
  }
  template <class A, class T>
  static hipError_t obstacles_t(const LaunchArgs& la, const ObstaclesArgs& a) {
    const long blocks = (la.B + kBlock - 1) / kBlock;
    const dim3 grid((unsigned)(blocks < kObstaclesMaxBlocks ? blocks : kObstaclesMaxBlocks));
    const ObsP<T>& P = *static_cast<const ObsP<T>*>(a.P);
    if constexpr (obstacles_use_lds<A, T>()) {
      static const bool off = measurement_env("ABRK_OBS_PLAIN") != nullptr;  // measurement switch: the one-pass kernel
      if (!off && P.n * (A::N - 2) <= 64) {
        hipLaunchKernelGGL((obstacles_lds_kernel<A, T>), grid, dim3(kBlock), 0, la.stream, Launch<A, T>::arm_of(la), P, la.B,
                           (const T*)a.q, (T*)a.u, a.acc);
        return hipGetLastError();
      }
    }
    hipLaunchKernelGGL((obstacles_kernel<A, T>), grid, dim3(kBlock), 0, la.stream, Launch<A, T>::arm_of(la), P, la.B,
                       (const T*)a.q, (T*)a.u, a.acc);
    return hipGetLastError();
  }
  static hipError_t obstacles(int dt, const LaunchArgs& la, const ObstaclesArgs& a) {
    return dt == 0 ? obstacles_t<AD, double>(la, a) : obstacles_t<AF, float>(la, a);
  }
  static const ArmOps* ops() {
    static const ArmOps o = {AD::N, &dyn, &osc, &sliding, &joint, AD::N == 2 ? &rollout : nullptr, &ik,
                             &floating, &obstacles};
    return &o;
  }
};

}  // namespace abrk
