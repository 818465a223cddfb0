// post_seq.hpp -- the order of the phases of post_core.hpp for one candidate.
// `Ex` supplies "run this phase on every thread of the workgroup, then barrier" (`par`)
// (`single`: on thread 0 only, no barrier) and the thread count (`nthreads`): DevExec in payne_hip.hip (threadIdx + __syncthreads),
// HostExec in tests/emul/cpu_emul.cpp.
//
// LOG2N > 0 selects the compile-time FFT geometry for spectra of exactly 2^LOG2N points
// processed by NT threads (strides, trip counts and twiddle steps become immediates);
// LOG2N = 0 is the runtime-geometry path (any size, any thread count).  A candidate whose
// R-stage window needs a shorter FFT than the vsini stage falls back to the runtime path
// for that stage only.
#pragma once
#include "post_core.hpp"

#ifdef __HIPCC__
#define PAYNE_SEQ __device__ __forceinline__
#define PAYNE_SEQ_CALL __device__ __attribute__((noinline))
#else
#define PAYNE_SEQ inline
#define PAYNE_SEQ_CALL inline
#endif

namespace payne {

// scratch layout (doubles): [0, nthr) chi^2 partials | [nthr, nthr + nthr/2) mask counts (ints) | result
PAYNE_HD int scratch_doubles(int nthr) { return nthr + nthr / 2 + 2; }

template <class Ex>
PAYNE_SEQ c32* fft_run_tiled(Ex& ex, c32* a_, c32* b_, int M, const c32* tw_, int tw_n, bool conj_last, c32* tile);
// M-point complex FFT by ping-pong between a and b, runtime geometry; returns where the result is.
template <class Ex>
PAYNE_SEQ c32* fft_run(Ex& ex, c32* a, c32* b, int M, const c32* tw, int tw_n, bool conj_last) {
  if (c32* tile = ex.tile())
    if (fft_tiled_ok(M)) return fft_run_tiled(ex, a, b, M, tw, tw_n, conj_last, tile);
  c32 *src = a, *dst = b;
  int p = 1;
  while (p < M) {
    const int R = pass_radix(M, p);
    const bool cj = conj_last && (p * R == M);
    if (R == 8) ex.par([&](int t, int n) { fft_pass<8>(t, n, src, dst, M, p, tw, tw_n, cj); });
    else if (R == 4) ex.par([&](int t, int n) { fft_pass<4>(t, n, src, dst, M, p, tw, tw_n, cj); });
    else ex.par([&](int t, int n) { fft_pass<2>(t, n, src, dst, M, p, tw, tw_n, cj); });
    c32* t_ = src; src = dst; dst = t_;
    p *= R;
  }
  return src;
}

// Four-step form (fft4_* in post_core.hpp) for executors that own an LDS tile (`ex.tile()`): two round trips
// through the global workspace per transform.  The closing pass of one tile and the opening pass of the next
// touch different buffers and share a barrier interval.
//   a -> b (step 1), b -> a (step 2); the result is in a.
template <class Ex>
PAYNE_SEQ c32* fft_run_tiled(Ex& ex, c32* a_, c32* b_, int M, const c32* tw_, int tw_n, bool conj_last, c32* tile) {
  auto a = Ex::buf(a_);
  auto b = Ex::buf(b_);
  auto tw = Ex::twid(tw_);
  auto X = Ex::lds(tile);
  auto Y = Ex::lds(tile + fft_tile_complex());
  const int B = M / kTileA;
  // step 1: a -> b
  ex.par([&](int t, int n) { fft4_s1_load(t, n, a, X, B, 0); });
  for (int c0 = 0; c0 < B; c0 += kTileC) {
    ex.par([&](int t, int n) { fft4_s1_mid(t, n, X, Y, tw, tw_n); });
    const bool more = c0 + kTileC < B;
    ex.par([&](int t, int n) {
      fft4_s1_store(t, n, Y, b, tw, tw_n, ColRun{c0});
      if (more) fft4_s1_load(t, n, a, X, B, c0 + kTileC);
    });
  }
  // step 2: b -> a.  The tile buffers alternate, so the closing pass of a tile (which reads one buffer) and the
  // opening pass of the next (which fills the other) share a barrier interval as well.
  auto P = X, Q = Y;
  ex.par([&](int t, int n) { fft4_s2_load(t, n, b, P, B, 0, tw, tw_n); });
  for (int k0 = 0; k0 < kTileA; k0 += kTileK) {
    const bool more = k0 + kTileK < kTileA;
    if (B == 128) {                                       // 8 x 8 x 2: P -> Q -> global, next tile into P
      ex.par([&](int t, int n) { fft4_s2_pass<8, false>(t, n, P, Q, a, B, 8, k0, tw, tw_n, false); });
      ex.par([&](int t, int n) {
        fft4_s2_pass<2, true>(t, n, Q, P, a, B, 64, k0, tw, tw_n, conj_last);
        if (more) fft4_s2_load(t, n, b, P, B, k0 + kTileK, tw, tw_n);
      });
    } else {                                              // 8 x 4 or 8 x 8: P -> global, next tile into Q
      ex.par([&](int t, int n) {
        if (B == 32) fft4_s2_pass<4, true>(t, n, P, Q, a, B, 8, k0, tw, tw_n, conj_last);
        else fft4_s2_pass<8, true>(t, n, P, Q, a, B, 8, k0, tw, tw_n, conj_last);
        if (more) fft4_s2_load(t, n, b, Q, B, k0 + kTileK, tw, tw_n);
      });
      auto t_ = P; P = Q; Q = t_;
    }
  }
  return a_;
}

// The same with compile-time geometry (M points, NT threads, pass-ordered twiddles `twf`).
template <int M, int P, int NT, class Ex, class BP, class TP>
PAYNE_SEQ BP fft_fixed_passes(Ex& ex, BP src, BP dst, TP twf, unsigned sign_last, bool edge) {
  if constexpr (P >= M) {
    return src;
  } else {
    constexpr int R = plan_radix(M, P);
    constexpr bool last = (P * R >= M);
    const unsigned sign = last ? sign_last : 0u;
    ex.par([&](int t, int) { fft_pass_fixed<R, M, P, NT>(t, src, dst, twf, sign, last && edge); });
    return fft_fixed_passes<M, P * R, NT>(ex, dst, src, twf, sign_last, edge);
  }
}
// One body for the four transforms of a candidate (2 stages x forward/inverse): kept out of
// line so the instruction stream stays small enough for the instruction cache.  Ex::buf /
// Ex::twid put the address space of the buffers and of the twiddle table into the pointer types.
template <int M, int NT, class Ex>
PAYNE_SEQ_CALL c32* fft_fixed(Ex& ex, c32* src, c32* dst, const c32* twf, unsigned sign_last, bool edge) {
  auto s = Ex::buf(src);
  auto d = Ex::buf(dst);
  return fft_fixed_passes<M, 1, NT>(ex, s, d, Ex::twid(twf), sign_last, edge) == s ? src : dst;
}

// Executors whose buffers are LDS say so through this trait: the LAST inverse transform of a likelihood evaluation and the
// observed-grid loop behind it then run as ONE phase (inverse_and_obs below).
template <class Ex> struct ex_lds_tail { static constexpr bool value = false; };
#ifdef __HIP_DEVICE_COMPILE__
// a barrier that waits for the wave's LDS traffic only: global loads stay in flight across it (__syncthreads() drains them)
__device__ __forceinline__ void post_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int M, int P, int NT, class BP, class TP>
__device__ __forceinline__ BP fft_fixed_passes_lds(int tid, BP src, BP dst, TP twf, unsigned sign_last) {
  if constexpr (P >= M) {
    return src;
  } else {
    constexpr int R = plan_radix(M, P);
    constexpr bool last = (P * R >= M);
    fft_pass_fixed<R, M, P, NT>(tid, src, dst, twf, last ? sign_last : 0u, false);
    post_lds_barrier();
    return fft_fixed_passes_lds<M, P * R, NT>(tid, dst, src, twf, sign_last);
  }
}
// The tail of a likelihood evaluation as ONE phase: the records of the observed pixels (16 bytes a pixel, 58 KB per candidate at
// C2 -- the phase that reads them is bound by those bytes at what an XCD's L2 hands a CU) are REQUESTED first and stay in flight
// under the instrumental stage's whole inverse transform, whose passes are inlined here behind LDS-only barriers (a call, or
// __syncthreads(), waits for every outstanding load); the loop then finds them in registers.  z: the tapered spectrum (Y of
// rfft_taper_phase), zo: the other buffer.  Returns the thread's chi^2 partial.
template <int M, int NT, int OU, class Ex>
__device__ __forceinline__ double inverse_and_obs(int tid, const PostTables& T, const CandState& S, const Window& W, const c32* twf,
                                                  c32* z, c32* zo) {
  ObsRec rec[OU];
  obs_fast_issue<OU>(NT, T, tid, rec);
  auto r = fft_fixed_passes_lds<M, 1, NT>(tid, Ex::buf(z), Ex::buf(zo), Ex::twid(twf), 0x80000000u);
  const float* conv = (r == Ex::buf(z)) ? (const float*)z : (const float*)zo;
  return phase_obs<OU>(tid, NT, T, S, W, conv, nullptr, -1, false, &rec);
}
#endif

// Executors that keep a 65 536-point stage on the compute unit (post_onchip.hpp: the spectrum in registers, LDS as the
// transpose buffer) say so through this trait; their convolution stage is chip_conv_stage (post_kernels.hpp).
constexpr int kChipN1 = 65536;
template <class Ex> struct ex_chip { static constexpr bool value = false; };
template <bool VSINI, class Ex>
PAYNE_SEQ float* chip_conv_stage(Ex& ex, float* work, const TaperArgs& ta, bool& edge, const float* src0, const Window* rs, bool zin, bool scrub);

// One real FFT-convolution stage of n points sitting in `work` (other buffer: `other`).
// `twf`: pass-ordered table of the fixed geometry (LDS or global); T.tw: plain full circle.
// `edge` in: the caller wants spec[0]=spec[1], spec[-1]=spec[-2] applied to the result;
// out: whether that is still to be done (the fixed-geometry transform does it in its last pass).
// `rs` (chip executors only): the stage's input is `src0` resampled through this window (the resampling phase was skipped).
template <int LOG2N, int NT, bool VSINI, class Ex>
PAYNE_SEQ float* conv_stage(Ex& ex, const PostTables& T, const c32* twf, float* work, float* other, int n,
                            const TaperArgs& ta, bool& edge, const float* src0 = nullptr, const Window* rs = nullptr,
                            bool have_y = false,     // have_y: `work` already holds the tapered transform (T.raw_freq rows: slots_commit)
                            bool zin = false, bool scrub = true,     // chip executors: src0 is the row's transform (chip_layout); NaN -> 0 on the way in
                            bool defer_inverse = false) {   // fixed geometry of exactly n points: leave the inverse transform to the caller (returns the tapered spectrum)
  const int M = n / 2;
  if constexpr (ex_chip<Ex>::value) {
    if (n == kChipN1) return chip_conv_stage<VSINI>(ex, work, ta, edge, src0, rs, zin, scrub);
  }
  if constexpr (LOG2N > 0) {
    constexpr int MF = (1 << LOG2N) / 2;
    if (M == MF) {
      c32* z = have_y ? (c32*)work : fft_fixed<MF, NT>(ex, (c32*)work, (c32*)other, twf, 0u, false);
      constexpr int PU = unroll_for((1 << LOG2N) / NT) / 4;
      if (!have_y) ex.par([&](int t, int) { rfft_taper_phase<VSINI, PU>(t, NT, Ex::buf(z), MF, Ex::twid(twf + plan_total(MF)), 1, ta); });
      c32* zo = ((float*)z == work) ? (c32*)other : (c32*)work;
      if (defer_inverse) return (float*)z;
      float* res = (float*)fft_fixed<MF, NT>(ex, z, zo, twf, 0x80000000u, edge);
      edge = false;
      return res;
    }
  }
  const c32* tw = T.tw;
  c32* z = fft_run(ex, (c32*)work, (c32*)other, M, tw, T.nmax, false);
  ex.par([&](int t, int nt) { rfft_taper_phase<VSINI>(t, nt, z, M, tw, T.nmax / (2 * M), ta); });
  c32* zo = ((float*)z == work) ? (c32*)other : (c32*)work;
  return (float*)fft_run(ex, z, zo, M, tw, T.nmax, true);
}

// out_stage: -1 chi^2 only | 0 raw ANN | 1 after vsini | 2 getspec on obs grid | 3 genspec (x blaze) | 5 after vsini, shifted | 6 after vsini, without the spec[0]=spec[1] edge rule
//            | 7 after vsini, interpolated from the stage's own resampled grid onto the OBSERVED grid (NaN outside): smoothspec('vsini', outwave=...)
// `early`: called by every thread in the first phase once ALL of the phase's global loads have been requested (the row, the
// record, theta) and before any of them is waited for -- the caller's own start-up traffic (the LDS kernel's twiddle table and
// photometric terms, requested before this function) is committed there, so its round trip and the row's are ONE round trip.
struct NoEarly { PAYNE_HD void operator()() const {} };
template <int LOG2N, int NT, class Ex, class Early = NoEarly>
PAYNE_SEQ void run_candidate(Ex& ex, const PostTables& T, const c32* twf, const double* th, double instr_factor,
                             const float* raw, float* bufA, float* bufB, CandState& S, double* red,
                             float* out, int out_stage, double* chi2_out, const CandState* prep = nullptr, Early early = Early(),
                             int raw_freq = -1) {                     // (-1: what the tables say; the LDS kernel passes what its launch resolved to)
  const bool rawf = raw_freq < 0 ? T.raw_freq != 0 : raw_freq != 0;
  // identity vsini maps: the row goes (NaN-scrubbed) straight to the FFT buffer.  The test reads theta: with the plain
  // executors it is made AFTER the row has been requested (a global load and its wait ahead of that request was a round trip
  // of its own at the start of every workgroup); only an on-chip stage, which reads the row itself, needs it before.
  const bool maybe_direct = (out_stage != 0) && T.rot_identity;
  // rows handed over in the frequency domain (the output layer carried the forward transform): every candidate starts at the
  // taper, applied on the way from global memory to LDS (slots_issue / slots_commit) -- one that does not rotate with the taper
  // of u = 0, which is 1 in every bin (and without the NaN scrub of the rotating branch: a row is all NaN or not at all, and a
  // NaN row stays NaN through the transform back)
  // (the on-chip stage of a 65 536-point spectrum takes the transformed row straight into its registers: chip_conv)
  const bool freq_chip = ex_chip<Ex>::value && T.n1 == kChipN1 && rawf;
  const bool freq = (LOG2N > 0 && rawf) || freq_chip;
  constexpr int MFq = LOG2N > 0 ? (1 << LOG2N) / 2 : 4;
  constexpr int SU = (MFq / 2 + NT - 1) / NT;          // slots per thread (2 at 4096 points on 512 threads)
  // per-pixel loops: LOG2N > 0 knows the pixels per thread (4096 / 512 = 8); the general path unrolls by 16
  constexpr int UX = LOG2N > 0 ? unroll_for((1 << LOG2N) / NT) : (NT >= 1024 ? 8 : 16);   // (1024 threads: 128 registers each)
  // an on-chip stage reads the row itself
  const bool may_fuse = maybe_direct && row_vectorised(T.npix, raw) && ex_chip<Ex>::value && T.n1 == kChipN1;
  bool direct = may_fuse ? (th[5] != 0.0) : false;
  const bool fused_row = may_fuse && (direct || freq_chip);
  ex.par([&](int t, int n) {
    RowRegsT<UX / 4> row;
    SlotRegs<SU> slots;
    // (A kernel that can take either kind of row requests PIXEL rows late, in the commit below: with both requests in this block the
    //  compiler's wait-count pass, which merges what may be in flight over both paths, saw the pixel path's destination registers as
    //  pending in the OTHER path and put `s_waitcnt vmcnt(0)` in front of the slots' requests -- the kernel's own table first, THEN
    //  the row: a second memory round trip at the start of every workgroup, 0.5 us of the C2 post kernel since round 4.)
    constexpr bool kLatePixels = LOG2N > 0;
    if (freq && !freq_chip) slots_issue<SU>(t, NT, MFq, raw, T.twf + plan_total(MFq), slots);
    else if (!fused_row && !kLatePixels) phase_load_issue(t, n, T.npix, raw, row);          // in flight during the setup chains
    PrepRegs pr;
    if (prep) phase_take_prep_issue(t, prep, pr);      // per-candidate scalars were computed ahead of the kernel
    double th5 = 0.0;
    if (!may_fuse && (maybe_direct || freq)) th5 = th[5];   // (requested with the others; looked at below)
    early();
    if (!may_fuse) direct = maybe_direct && (th5 != 0.0);
    if (prep) phase_take_prep_commit(t, pr, S);
    else phase_setup(t, n, T, th, instr_factor, S);
    ex.mark(128);                                      // (diagnostic build: end of the instrument / mask-probe chain)
    if (freq && !freq_chip) {
      if constexpr (Ex::kTwLds && LOG2N > 0) {         // the split factors this thread loaded for its slots complete the kernel's LDS table
        auto tl = Ex::buf(const_cast<c32*>(twf) + plan_total(MFq));
#pragma unroll
        for (int q = 0; q < SU; ++q) { const int j = t + q * NT; if (j < MFq / 2) stc(tl, j, slots.w[q]); }
      }
      // (NaN -> 0 is the rotating branch's nan_to_num; with resampling maps -- rows of a resampled grid -- every candidate of the
      //  batch rotates, or the launch would have been switched to pixels: PostArgs::rot_flag)
      slots_commit<SU>(t, NT, MFq, slots, Ex::buf((c32*)bufB), vsini_taper_args(T, th5), th5 != 0.0);
    }
    else if (!fused_row) {
      if (kLatePixels) phase_load_issue(t, n, T.npix, raw, row);
      phase_load_commit(t, n, T.npix, raw, row, direct ? bufB : bufA, direct);
    }
  });
  float* spec = bufA;
  float* work = bufB;
  if (out_stage == 0) {
    ex.par([&](int t, int n) { for (int i = t; i < T.npix; i += n) out[i] = spec[i] + kBase; });
    return;
  }
  const bool rot = S.do_rot != 0, smooth = S.do_smooth != 0;
  bool edges_pending = false;
  if (rot || freq) {
    if (!direct && !freq) ex.par([&](int t, int n) { phase_rot_resample(t, n, T, spec, work); });
    TaperArgs ta{};
    ta.vs_tab = T.vs_tab; ta.vs_tab_n = T.vs_tab_n;
    ta.vs_c = rot ? S.vs_a * T.vs_val : 0.0;           // u_k = 2 pi sigma k/(n dv)   (smoothing.py:612-614)   [freq: applied already]
    ta.vs_c64 = ta.vs_c * (1.0 / kVsTabStep);
    // identity maps: the convolved buffer IS the spectrum on the ANN grid (npix == n1), and the
    // transform's last pass can apply the edge rule itself
    bool edge = rot && T.rot_identity != 0 && out_stage != 6 && out_stage != 7;   // 6, 7 = smoothspec('vsini') itself: no edge rule
    float* conv = conv_stage<LOG2N, NT, true>(ex, T, twf, work, spec, T.n1, ta, edge, fused_row ? raw : nullptr, nullptr, freq && !freq_chip,
                                              freq_chip, direct);
    if (rot && out_stage == 7) {
      // np.interp(outwave, w_resampled, conv, left = right = NaN) (smoothing.py:308-311): the resampled grid is resample_wave of the
      // WHOLE model grid -- the window of "no mask, no shift"
      const Window Wv = window_from_counts(T, 0.0, 0.0, 0, T.npix);
      ex.par([&](int t, int n) { store_partial(t, phase_obs<UX>(t, n, T, S, Wv, conv, out, 2, true), red); });
      return;
    }
    float* dst = (conv == bufA) ? bufB : bufA;
    if (T.rot_identity) { float* t_ = dst; dst = conv; conv = t_; }
    else {
      const bool er = rot && out_stage != 6 && T.npix >= 4;          // the edge rule rides along
      ex.par([&](int t, int n) { phase_rot_back<UX>(t, n, T, conv, dst, er); });
      edge = rot && out_stage != 6 && !er;
    }
    spec = dst;
    work = conv;
    edges_pending = edge;
    if (edges_pending && (out_stage == 1 || out_stage == 5 || !smooth)) {
      ex.par([&](int t, int) { phase_rot_edges(t, T.npix, spec); });
      edges_pending = false;
    }
  }
  if (out_stage == 7) {                                       // (no rotation: the reference's taper is 0 / 0 there -- NaN everywhere)
    ex.par([&](int t, int n) { for (int i = t; i < T.nobs; i += n) out[i] = nanf_(); });
    return;
  }
  if (out_stage == 1 || out_stage == 5 || out_stage == 6) {   // 5: still shifted by -1 (input of the LSF kernel); 6: no edge rule
    const float base = out_stage == 5 ? 0.f : kBase;
    ex.par([&](int t, int n) { for (int i = t; i < T.npix; i += n) out[i] = spec[i] + base; });
    return;
  }
  const float* on_grid = spec;
  // (executors whose stages are CALLS that use the whole register file -- the on-chip stages -- read the window from the record
  //  in LDS wherever the record holds it: twenty-five registers a thread less to carry across those calls)
  Window Wl{};
  const Window* Wp = &Wl;
  bool tail_done = false;
  if (smooth) {
    if (S.win_ready) {                                 // mask counts known since setup
      if (edges_pending) ex.par([&](int t, int) { phase_rot_edges(t, T.npix, spec); });
      if (S.w_ready) { if constexpr (ex_chip<Ex>::value) Wp = &S.W; else Wl = S.W; }   // ... and so is the window (prep_candidate)
      else Wl = window_from_counts(T, S.dop, S.g_a, S.win_below, S.win_notabove);   // by every thread
      ex.mark(0);
    } else {
      const int nthr = ex.nthreads();
      int* cnt = reinterpret_cast<int*>(red + nthr);
      ex.par([&](int t, int n) {
        if (edges_pending) phase_rot_edges(t, T.npix, spec);   // the count does not touch the spectrum
        phase_mask_count(t, n, T, S, cnt);
      });
      Wl = make_window(T, S, cnt, n_slots(nthr));
    }
    const Window& W = *Wp;
    if (!W.bad) {
      // (an executor that keeps the stage on the compute unit gathers the resampled points while it loads them)
      const bool gather = ex_chip<Ex>::value && T.geo && W.n2 == kChipN1;
      if (!gather) ex.par([&](int t, int n) { phase_R_resample<UX>(t, n, T, S, W, spec, work); });
      TaperArgs ta{};
      ta.g_c2 = W.g_c2;
      bool no_edge = false;
      // (a stage that gathers its input itself has all of it in registers before it stores anything: its output goes where its
      //  input was -- one 256 KB buffer per workgroup in flight instead of two, 67 MB for the 256 workgroups of a C5 launch)
      // chi^2 alone from a window of the fixed geometry's own length, whole blocks of records: the inverse transform and the
      // observed-grid loop as one phase, the records requested ahead of the transform (inverse_and_obs)
      bool fused_tail = false;
      if constexpr (ex_lds_tail<Ex>::value && LOG2N > 0 && UX <= 8) {    // (a thread's block of records stays in registers: 4 x UX of them)
        fused_tail = !gather && out == nullptr && T.obs_f1 != nullptr && T.npoly == 0 && W.n2 == (1 << LOG2N) && obs_fast_ok(T, UX * NT);
      }
      on_grid = conv_stage<LOG2N, NT, false>(ex, T, twf, gather ? spec : work, spec, W.n2, ta, no_edge, gather ? spec : nullptr, gather ? &W : nullptr,
                                             false, false, true, fused_tail);
#ifdef __HIP_DEVICE_COMPILE__
      if constexpr (ex_lds_tail<Ex>::value && LOG2N > 0 && UX <= 8) {
        if (fused_tail) {
          c32* z = (c32*)const_cast<float*>(on_grid);
          c32* zo = (on_grid == work) ? (c32*)spec : (c32*)work;
          ex.par([&](int t, int) { store_partial(t, inverse_and_obs<(1 << LOG2N) / 2, NT, UX, Ex>(t, T, S, W, twf, z, zo), red); });
          tail_done = true;
        }
      }
#endif
    }
  }
  const Window& W = *Wp;
  if (!tail_done) ex.par([&](int t, int n) { store_partial(t, phase_obs<UX>(t, n, T, S, W, on_grid, out, out_stage), red); });
  // the sum of the per-wave partials: thread 0 alone, no closing barrier (it is also the only reader)
  ex.single([&](int n) {
    double s = 0.0;
    const int ns = n_slots(n);
    for (int i = 0; i < ns; ++i) s += red[i];
    *chi2_out = s;
  });
}

}  // namespace payne
