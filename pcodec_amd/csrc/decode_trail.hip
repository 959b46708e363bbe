// decode_trail.hip -- the expanders that run UNDER the tANS walk.
//
// dec_walk_kernel is a latency chain: one wave per SIMD, the LDS full of tANS tables, 6.5 ms per launch whatever the launch holds, the
// SIMDs more than half idle.  dec_expand_kernel then streamed for another 6 ms.  Two kernels from two streams do share the CUs
// (scripts/micro/overlap_test.hip: a 6.6 ms LDS chase and a 6.1 ms streaming kernel finish together after 6.9 ms), so the expansion of
// the common chunks now runs on a second stream WHILE the walk goes on, in a kernel that needs no LDS at all:
//   * grid = one block of four waves per walker block (persistent: at most four blocks per CU, later walker blocks in later rounds); wave v
//     follows chunk slots 2v and 2v + 1 of its walker block, batch after batch, as the walker publishes them (progress words in global
//     memory, 1 + completed batches; scripts/micro/handoff_test.hip is the hand-over on its own).  96 VGPRs a wave: four of them and the
//     walker's 128 are a SIMD's 512;
//   * it takes the chunks of ONE latent variable with 2..64 bins and offsets of up to 16 bits (classic mode without lookback -- the walker
//     marks them): the state of eight chunks per SIMD has to fit the registers the walker leaves, and a second variable does not;
//   * the bins live in registers, a bin per lane: bin lookup = ds_bpermute (the LDS crossbar, no LDS memory);
//   * a batch's offset section is fetched with coalesced requests (lane l: dword l from the dword the section starts in) and a lane gets
//     the three dwords its four fields lie in through the crossbar (a lane loading 12 bytes at its own bit position is 64 addresses an
//     instruction: the address unit took ~700 cycles over each);
//   * the delta moments are wave-uniform registers (one wave owns a chunk from its first batch to its last: no turn-taking);
//   * what the walker wrote is read with agent-scope loads (the two kernels may sit on different XCDs, whose L2s are not coherent with
//     one another for plain accesses); the numbers leave as contiguous non-temporal KBs (store_u64_batch, decode_fast.hip).
// A wave stays one batch behind the walker where it can, so that the symbols and section start of the batch it expands were requested
// an iteration earlier; its two chunks are expanded in lockstep -- in the steady state as straight-line code (trail_fast_pair).  Loads are
// unconditional and into registers nothing else writes: a register zeroed first and loaded under a condition makes the compiler wait, at
// the zeroing, for every load but the last few, whatever is really in flight.  Every wait is bounded: a wave that sees no progress for
// about 55 ms gives its chunks back and dec_expand_kernel expands them afterwards: ownership is explicit -- an expander wave marks every chunk it
// has expanded to the end in the walker block's progress line (word kTrailDoneWord + slot), and dec_expand_kernel, which runs when both kernels
// have ended, takes every fused chunk WITHOUT that mark (the walker started late, the expander timed out at any point, or the expander kernel
// refused the chunk) and counts it (pco_gfx_trail_givebacks).
// Measured (8192 chunks of 2^18 u64, delta 1): walk 6.5 + expand 6.2 ms back to back -> 10.6 ms together (the publishing walker alone 7.2,
// these expanders alone 5.6; without the expanders' output stream the pair takes 9.2: what is left is the two kernels' contention for the
// memory system and the issue slots, not the hand-over -- a build that ignores the progress words finishes in 10.0).
// Reference semantics: page_latent_decompressor.rs:15-44,89-213, delta/consecutive.rs:35-50, mode/classic.rs:14-24 (as dec_expand_kernel).
#include "decode_fast.hip"

namespace pcogfx {

constexpr uint32_t kTrailWaves = 4;            // per walker block
constexpr uint32_t kTrailSlotsPerWave = 2;     // wave v: chunk slots 2 v, 2 v + 1
constexpr uint32_t kTrailSpinLimit = 1u << 14; // polls (~3.4 us apart, ~55 ms) without progress before a wave gives up.  Giving up is always safe since round 5: a chunk is
                                               // skipped by dec_expand_kernel only when its expander wave has marked it DONE (kTrailDoneWord), so the limit only bounds how long
                                               // a call can stall when the walker is not running beside the expanders (a device shared with another process)
// DecPlan as 64 words (the expanders fetch it with one request, a word per lane)
constexpr uint32_t kPlanN = 1, kPlanModeKind = 2, kPlanModeK = 3, kPlanModeBase = 4, kPlanNumKind = 6, kPlanPresent = 8, kPlanNBins = 11, kPlanMaxOb = 14,
                   kPlanDeltaKind = 17, kPlanDeltaOrder = 20, kPlanNlps = 23, kPlanMoments = 28, kPlanFused = 62;
static_assert(sizeof(DecPlan) == 256 && offsetof(DecPlan, n) == 4 * kPlanN && offsetof(DecPlan, mode_base) == 4 * kPlanModeBase && offsetof(DecPlan, num_kind) == 4 * kPlanNumKind &&
              offsetof(DecPlan, present) == 4 * kPlanPresent && offsetof(DecPlan, n_bins) == 4 * kPlanNBins && offsetof(DecPlan, max_ob) == 4 * kPlanMaxOb &&
              offsetof(DecPlan, delta_kind) == 4 * kPlanDeltaKind && offsetof(DecPlan, delta_order) == 4 * kPlanDeltaOrder && offsetof(DecPlan, nlps) == 4 * kPlanNlps &&
              offsetof(DecPlan, moments) == 4 * kPlanMoments && offsetof(DecPlan, fused) == 4 * kPlanFused, "DecPlan word layout");

#ifdef PCO_TRAIL_PLAINLD   // (measurement builds only: what the agent scope of the loads costs)
__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) { return *(const volatile uint32_t*)p; }
__device__ __forceinline__ uint64_t ld_agent(const uint64_t* p) { return *(const volatile uint64_t*)p; }
#else
__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint64_t ld_agent(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif
__device__ __forceinline__ uint32_t bperm(uint32_t byte_index, uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)byte_index, (int)v); }
template <class T> __device__ __forceinline__ T bperm_t(uint32_t byte_index, T v) {
  if constexpr (sizeof(T) == 8) return (T)(((uint64_t)bperm(byte_index, (uint32_t)((uint64_t)v >> 32)) << 32) | bperm(byte_index, (uint32_t)v));
  else return (T)bperm(byte_index, (uint32_t)v);
}

// (kept narrow on purpose: two chunks' worth of this live in a wave's scalar registers across the whole loop -- the flags are bits of one word
//  and the small fields one packed word; a bool each is a register pair, and the overflow went through v_readlane / scratch in the hot loop)
constexpr uint32_t kFlLive = 1, kFlHave = 2, kFlFast = 4, kFlSecValid = 8, kFlTwo = 16, kFlFast2 = 32, kFlTriv2 = 64;
template <class L> struct TrailChunk {
  uint32_t fl;                         // kFlLive; kFlHave: pf_* hold batch `next`; kFlFast: the common shape -- several bins, offsets of 1..7 bits (a full batch's section fits
                                       // one register across the wave), aligned output; kFlSecValid: batch `next`'s section(s) were requested at the end of the batch before;
                                       // kFlTwo / kFlFast2 (dec_trail_kernel<L, true>): two latent variables / in the common shape (primary offsets up to 15 bits, secondary up to 7)
  uint32_t shape;                      // num_kind (2 bits) | n_bins (7) | max_ob (5) | delta order (2) | mode kind (3) | secondary n_bins (7) | secondary max_ob (5)
  uint32_t ti, n, n_batches, next;
  gcptr_u8 src; uint64_t src_len; L PCO_GLOBAL* dst;
  L tbl_low; uint32_t tbl_ob;          // lane b: lower and offset bits of bin b
  L mom[2];                            // delta moments (wave-uniform)
  uint32_t pf_syms, pf_start;          // requested for batch `next` (a section starts less than 2^32 bits into its chunk)
  uint32_t start_live;                 // the section start of the batch between its stages A and B (pf_start is requested anew in between)
  uint32_t sec;                        // fast path: batch `next`'s section (dword `lane` from the dword it starts in)
  // dec_trail_kernel<L, true> only: the chunk's secondary variable (int-mult / float-mult / float-quant; never delta'd here) and what joins the two
  uint32_t mode_k; L mode_base;
  L tbl_low2; uint32_t tbl_ob2;
  uint32_t pf_syms2, pf_start2, start_live2;
  uint32_t sec_hi, sec_b;              // fast path of two-variable chunks: dwords 64 + lane of the primary's section, dword lane of the secondary's
  __device__ __forceinline__ bool live() const { return (fl & kFlLive) != 0; }
  __device__ __forceinline__ bool have() const { return (fl & kFlHave) != 0; }
  __device__ __forceinline__ bool fast_ok() const { return (fl & kFlFast) != 0; }
  __device__ __forceinline__ bool sec_valid() const { return (fl & kFlSecValid) != 0; }
  __device__ __forceinline__ bool two() const { return (fl & kFlTwo) != 0; }
  __device__ __forceinline__ bool fast2_ok() const { return (fl & kFlFast2) != 0; }
  __device__ __forceinline__ bool triv2_ok() const { return (fl & kFlTriv2) != 0; }   // kFlTriv2: two variables in the common shape whose secondary is ONE bin without offset bits (a constant: exact decimals under float-mult)
  __device__ __forceinline__ void set(uint32_t bit, bool v) { fl = v ? (fl | bit) : (fl & ~bit); }
  __device__ __forceinline__ uint32_t num_kind() const { return shape & 3u; }
  __device__ __forceinline__ uint32_t n_bins() const { return (shape >> 2) & 127u; }
  __device__ __forceinline__ uint32_t max_ob() const { return (shape >> 9) & 31u; }
  __device__ __forceinline__ uint32_t dord() const { return (shape >> 14) & 3u; }
  __device__ __forceinline__ uint32_t nlps() const { return dord(); }   // (consecutive delta of order k keeps k latents as state: delta/consecutive.rs)
  __device__ __forceinline__ uint32_t mode_kind() const { return (shape >> 16) & 7u; }
  __device__ __forceinline__ uint32_t n_bins2() const { return (shape >> 19) & 127u; }
  __device__ __forceinline__ uint32_t max_ob2() const { return (shape >> 26) & 31u; }
  // the walker's output for this chunk: symbols (256 per batch) and section start per batch of the primary variable; the secondary's follow one stride on
  __device__ __forceinline__ const uint8_t* syms(const uint8_t* sym_area, uint64_t sym_stride) const { return sym_area + ((uint64_t)ti * 3 + 1) * sym_stride; }
  __device__ __forceinline__ const uint64_t* starts(const uint64_t* offpos_area, uint64_t offpos_stride) const { return offpos_area + ((uint64_t)ti * 3 + 1) * offpos_stride; }
};
template <class L> struct TrailVar { L tbl_low; uint32_t tbl_ob, n_bins, max_ob, pf_syms, start_live; };
template <class L> __device__ __forceinline__ TrailVar<L> trail_primary(const TrailChunk<L>& c) { return TrailVar<L>{c.tbl_low, c.tbl_ob, c.n_bins(), c.max_ob(), c.pf_syms, c.start_live}; }
template <class L> __device__ __forceinline__ TrailVar<L> trail_secondary(const TrailChunk<L>& c) { return TrailVar<L>{c.tbl_low2, c.tbl_ob2, c.n_bins2(), c.max_ob2(), c.pf_syms2, c.start_live2}; }
// what stage A of a batch leaves for stage B (kept small: both chunks' worth are live across the section loads).  syms: the lane's four bin
// symbols; obs: their offset-bit counts, a byte each; excl: bits of the section before this lane's first field; s*: the section itself for
// offsets of up to 16 bits, lane l of s_j holding its dword 64 j + l (counted from the dword the section starts in) -- fetched with one to
// three coalesced requests and handed to the lanes that need a dword through the crossbar.  (A lane loading the 12 bytes at its own bit
// position is one instruction and 64 addresses: the address unit took ~700 cycles over each, more than a CU has per batch.)  The lowers are
// looked up in stage B rather than carried.
struct TrailItem { uint32_t syms, obs, excl, s0, s1, s2; };

typedef uint32_t trail_u32x4 __attribute__((ext_vector_type(4)));
template <class L> __device__ __forceinline__ uint32_t trail_cnt(const TrailChunk<L>& c, uint32_t batch) {
  const uint32_t n_remaining = c.n - batch * kBatchN;
  const uint32_t rem = n_remaining > c.nlps() ? n_remaining - c.nlps() : 0;
  return rem < kBatchN ? rem : kBatchN;
}

// Request the walker's output for batch b of the chunk (addresses that depend on nothing the batch computes).  Unconditional, into
// registers nothing else writes: a register that is zeroed first and loaded under a condition makes the compiler wait, at the zeroing, for
// every load but the last few -- statically, whatever is really in flight (the section loads just issued, in this loop).  A batch without
// latents of its own (the tail of a delta'd chunk) reads its slot's stale bytes; every use is masked by the latent count.
struct TrailAreas { const uint8_t* sym_area; uint64_t sym_stride; const uint64_t* offpos_area; uint64_t offpos_stride; };
template <class L, bool kTwo = false> __device__ __forceinline__ void trail_request(TrailChunk<L>& c, uint32_t b, const TrailAreas& ar) {
  const uint8_t* syms = c.syms(ar.sym_area, ar.sym_stride); const uint64_t* starts = c.starts(ar.offpos_area, ar.offpos_stride);
  c.pf_syms = ld_agent((const uint32_t*)(syms + (uint64_t)b * kBatchN) + lane_id());
  c.pf_start = ld_agent((const uint32_t*)(starts + b));
  if (kTwo && !c.triv2_ok()) {   // (the secondary variable's slots are the next ones of the task: stale bytes for a chunk without one, never used; a constant secondary has nothing to fetch)
    c.pf_syms2 = ld_agent((const uint32_t*)(syms + ar.sym_stride + (uint64_t)b * kBatchN) + lane_id());
    c.pf_start2 = ld_agent((const uint32_t*)(starts + ar.offpos_stride + b));
  }
  c.set(kFlHave, true);
}

// Stage A: symbols -> offset-bit counts (register table), their prefix over the wave, the window load issued.
template <class L> __device__ __forceinline__ void trail_stage_a(const TrailVar<L>& c, uint32_t cnt, TrailItem& it) {
  const uint32_t lane = lane_id();
  // the walker's layout has chain c, block b of a 64-symbol group at dword 4 c + b; this lane wants chain lane % 4 of block lane / 4
  const uint32_t mine = bperm(4u * ((lane & 48u) + 4u * (lane & 3u) + ((lane >> 2) & 3u)), c.pf_syms);
  it.syms = c.n_bins <= 1 ? 0u : quad_transpose_u8(mine, lane & 3);
  uint32_t t = 0; it.obs = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t looked = bperm(4u * ((it.syms >> (8 * k)) & 63u), c.tbl_ob);   // (by every lane: a lane that sat out would read as zero for the lanes that index it)
    const uint32_t o = 4 * lane + k < cnt ? looked : 0u;
    it.obs |= o << (8 * k); t += o;
  }
  it.excl = c.max_ob == 0 ? 0u : wave_incl_scan(t) - t;
}

// The section of a batch (offsets of up to 16 bits: at most 4096 bits + the dwords a lane's 64-bit window may reach into), requested as soon
// as its start is known: dword 64 j + lane of the section, counted from the dword it starts in.  Reads are clamped to the buffer's 16 bytes of slack.
template <class L> __device__ __forceinline__ void trail_section(const TrailVar<L>& c, gcptr_u8 src, uint64_t src_len, uint32_t cnt, TrailItem& it) {
  const uint32_t lane = lane_id();
  const uint32_t d0 = c.start_live >> 5, nd = c.max_ob <= 16 ? (((c.start_live & 31u) + cnt * c.max_ob + 31u) >> 5) + 2u : 0u;   // (uniform)
  const uint32_t last = (uint32_t)((src_len + 12) >> 2);
  auto dword = [&](uint32_t i) -> uint32_t { const uint32_t dw = d0 + i < last ? d0 + i : last; return load_u32_le(src + 4ull * dw); };
  it.s0 = dword(lane);
  if (nd > 64) it.s1 = dword(64 + lane);
  if (nd > 128) it.s2 = dword(128 + lane);
}

// Stage B: the lowers (register table), the offsets cut from the window, their sum.
template <class L> __device__ __forceinline__ void trail_stage_b(const TrailVar<L>& c, uint32_t cnt, const TrailItem& it, L out[4]) {
  const uint32_t lane = lane_id();
  L low[4];
#pragma unroll
  for (int k = 0; k < 4; k++) { const L lo = bperm_t<L>(4u * ((it.syms >> (8 * k)) & 63u), c.tbl_low); low[k] = 4 * lane + k < cnt ? lo : (L)0; }
  if (c.max_ob == 0) {
#pragma unroll
    for (int k = 0; k < 4; k++) out[k] = low[k];
    return;
  }
  uint64_t bit = (uint64_t)c.start_live + it.excl;
  if (c.max_ob <= 16) {
    // the lane's four fields span at most 64 bits from bit `rel` of the fetched section: dwords di, di + 1, di + 2 come out of the lanes that hold them
    const uint32_t rel = (c.start_live & 31u) + it.excl, di = rel >> 5, sh = rel & 31u;
    const uint32_t nd = (((c.start_live & 31u) + cnt * c.max_ob + 31u) >> 5) + 2u;   // (uniform, as in trail_section)
    uint32_t w0, w1, w2;
    if (nd <= 64) { w0 = bperm(4u * di, it.s0); w1 = bperm(4u * di + 4u, it.s0); w2 = bperm(4u * di + 8u, it.s0); }   // (indices past lane 63 wrap: only bits no field uses come from there)
    else {
      auto fetch = [&](uint32_t i) -> uint32_t {
        const uint32_t a = bperm(4u * (i & 63u), it.s0), b = bperm(4u * (i & 63u), it.s1), cc = bperm(4u * (i & 63u), it.s2);
        return i < 64 ? a : (i < 128 ? b : cc);
      };
      w0 = fetch(di); w1 = fetch(di + 1); w2 = fetch(di + 2);
    }
    uint64_t v64 = ((uint64_t)__builtin_amdgcn_alignbit(w2, w1, sh) << 32) | __builtin_amdgcn_alignbit(w1, w0, sh);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t o = (it.obs >> (8 * k)) & 0xffu;
      out[k] = (L)(low[k] + (L)__builtin_amdgcn_ubfe((uint32_t)v64, 0u, o));
      v64 >>= o;
    }
    return;
  }
  // (offsets beyond 16 bits never come here: the walker leaves those chunks to dec_expand_kernel)
#pragma unroll
  for (int k = 0; k < 4; k++) out[k] = low[k];
}

// consecutive_decode (decode_kernel.hip) with the moments in registers
template <class L> __device__ __forceinline__ void trail_delta(L x[4], uint32_t order, L mom[2]) {
#pragma unroll
  for (int k = 0; k < 4; k++) x[k] = (L)(x[k] + lmid<L>());
#pragma unroll
  for (int m = 1; m >= 0; m--) {
    if ((uint32_t)m >= order) continue;
    const L e1 = x[0], e2 = (L)(e1 + x[1]), e3 = (L)(e2 + x[2]), t = (L)(e3 + x[3]);
    const L incl = wave_incl_scan(t);
    const L base = (L)(mom[m] + (L)(incl - t));
    x[0] = base; x[1] = (L)(base + e1); x[2] = (L)(base + e2); x[3] = (L)(base + e3);
    mom[m] = (L)(mom[m] + wave_last(incl));
  }
}

// The steady state of a wave -- both its chunks have a full batch ready, in the common shape (TrailChunk::fast_ok) -- as straight-line code:
// no per-chunk branches, so that the two chunks' lookup / scan / fetch chains (each a string of crossbar and DPP round trips that nothing
// else in the wave covers) are scheduled into one another.  Returns the progress words requested for the next iteration.
template <class L>
__device__ __forceinline__ uint32_t trail_fast_pair(TrailChunk<L> (&S)[kTrailSlotsPerWave], const uint32_t (&ready)[kTrailSlotsPerWave], const uint32_t* pline, const TrailAreas& ar) {
  static_assert(kTrailSlotsPerWave == 2, "written for two chunks per wave");
  const uint32_t lane = lane_id();
  const uint32_t layout = 4u * ((lane & 48u) + 4u * (lane & 3u) + ((lane >> 2) & 3u));
  uint32_t syms[2], obs[2], excl[2];
  auto request_section = [&](TrailChunk<L>& c) {   // one register: at most 56 + 3 dwords; reads clamped into the buffer's 16 bytes of slack
    c.start_live = uni(c.pf_start);
    const uint32_t d0 = c.start_live >> 5, last = (uint32_t)((c.src_len + 12) >> 2);
    const uint32_t dw = d0 + lane < last ? d0 + lane : last;
    c.sec = load_u32_le(c.src + 4ull * dw);
  };
  // ---- stage A of both: the section if the last iteration has not asked for it already, symbols -> offset bits -> their prefix ----
#pragma unroll
  for (int q = 0; q < 2; q++) {
    TrailChunk<L>& c = S[q];
    if (!c.sec_valid()) request_section(c);
    syms[q] = quad_transpose_u8(bperm(layout, c.pf_syms), lane & 3);
    uint32_t t = 0, o4 = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { const uint32_t o = bperm(4u * ((syms[q] >> (8 * k)) & 63u), c.tbl_ob); o4 |= o << (8 * k); t += o; }
    obs[q] = o4; excl[q] = wave_incl_scan(t) - t;
  }
  // ---- behind the section loads: the next batches' requests, the next iteration's progress words ----
#pragma unroll
  for (int q = 0; q < 2; q++) {
    TrailChunk<L>& c = S[q];
    c.set(kFlHave, false);
    if (c.next + 1 < c.n_batches && ready[q] > c.next + 1) trail_request<L>(c, c.next + 1, ar);
  }
  const uint32_t pv_next = ld_agent(pline);
  // ---- stage B of both.  Before a batch's numbers are stored the NEXT batch's section is asked for: the memory counter of a wave is one
  // in-order queue of loads and stores, so a load issued behind the stores would be waited for together with their acknowledgement
  // (written-through output: microseconds) -- this way the stores have a whole iteration to drain before anything behind them is needed.
#pragma unroll
  for (int q = 0; q < 2; q++) {
    TrailChunk<L>& c = S[q];
    L x[4];
#pragma unroll
    for (int k = 0; k < 4; k++) x[k] = bperm_t<L>(4u * ((syms[q] >> (8 * k)) & 63u), c.tbl_low);
    const uint32_t rel = (c.start_live & 31u) + excl[q], di = rel >> 5, sh = rel & 31u;
    const uint32_t w0 = bperm(4u * di, c.sec), w1 = bperm(4u * di + 4u, c.sec), w2 = bperm(4u * di + 8u, c.sec);
    uint64_t v64 = ((uint64_t)__builtin_amdgcn_alignbit(w2, w1, sh) << 32) | __builtin_amdgcn_alignbit(w1, w0, sh);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t o = (obs[q] >> (8 * k)) & 0xffu;
      x[k] = (L)(x[k] + (L)__builtin_amdgcn_ubfe((uint32_t)v64, 0u, o));
      v64 >>= o;
    }
    if (c.dord()) trail_delta<L>(x, c.dord(), c.mom);
    L PCO_GLOBAL* o = c.dst + (uint64_t)c.next * kBatchN + 4 * lane;
    c.set(kFlSecValid, false);
    if (c.have() && c.n - (c.next + 1) * kBatchN >= kBatchN + c.nlps()) { request_section(c); c.set(kFlSecValid, true); }   // (a full batch follows and its start has been asked for)
#ifdef PCO_TRAIL_NOSTORE   // (measurement builds only)
    if ((x[0] ^ x[1] ^ x[2] ^ x[3]) == (L)0x9e3779b97f4a7c15ull)
#endif
    if constexpr (sizeof(L) == 8) {
#ifdef PCO_TRAIL_OLDSTORE
      typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
      u64x2 a; a.x = from_latent_ordered<L>(x[0], c.num_kind()); a.y = from_latent_ordered<L>(x[1], c.num_kind());
      u64x2 b; b.x = from_latent_ordered<L>(x[2], c.num_kind()); b.y = from_latent_ordered<L>(x[3], c.num_kind());
      ((u64x2 PCO_GLOBAL*)o)[0] = a; ((u64x2 PCO_GLOBAL*)o)[1] = b;
#else
      const unsigned long long y[4] = {from_latent_ordered<L>(x[0], c.num_kind()), from_latent_ordered<L>(x[1], c.num_kind()), from_latent_ordered<L>(x[2], c.num_kind()), from_latent_ordered<L>(x[3], c.num_kind())};
      store_u64_batch((unsigned long long PCO_GLOBAL*)(c.dst + (uint64_t)c.next * kBatchN), y);
#endif
    } else if constexpr (sizeof(L) == 4) {
      trail_u32x4 a; a.x = from_latent_ordered<L>(x[0], c.num_kind()); a.y = from_latent_ordered<L>(x[1], c.num_kind()); a.z = from_latent_ordered<L>(x[2], c.num_kind()); a.w = from_latent_ordered<L>(x[3], c.num_kind());
      __builtin_nontemporal_store(a, (trail_u32x4 PCO_GLOBAL*)o);
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) o[k] = from_latent_ordered<L>(x[k], c.num_kind());
    }
    c.next++;
  }
  return pv_next;
}


// join of a lane's four numbers with the (wave-uniform) mode decided once, not per number
template <class L> __device__ __forceinline__ void trail_join4(uint32_t mode_kind, uint32_t num_kind, L base, uint32_t mk, const L (&p)[4], const L (&q)[4], L (&out)[4]) {
  if (mode_kind == kFloatMult) {
#pragma unroll
    for (int k = 0; k < 4; k++) out[k] = join_one<L>(kFloatMult, num_kind, base, mk, p[k], q[k]);
  } else if (mode_kind == kIntMult) {
#pragma unroll
    for (int k = 0; k < 4; k++) out[k] = join_one<L>(kIntMult, num_kind, base, mk, p[k], q[k]);
  } else {
#pragma unroll
    for (int k = 0; k < 4; k++) out[k] = join_one<L>(kFloatQuant, num_kind, base, mk, p[k], q[k]);
  }
}

// trail_fast_pair for chunks of TWO latent variables (dec_trail_kernel<L, true>): both chunks have a full batch of both variables ready, the
// primary's offsets have up to 15 bits (section <= 123 dwords: two registers across the wave), the secondary's up to 7 (one register) -- float-mult
// decimals are 13 and 0.  Straight-line for the same reason: four unpackings' lookup / scan / fetch chains scheduled into one another.
template <class L>
__device__ __forceinline__ uint32_t trail_fast_pair2(TrailChunk<L> (&S)[kTrailSlotsPerWave], const uint32_t (&ready)[kTrailSlotsPerWave], const uint32_t* pline, const TrailAreas& ar) {
  static_assert(kTrailSlotsPerWave == 2, "written for two chunks per wave");
  const uint32_t lane = lane_id();
  const uint32_t layout = 4u * ((lane & 48u) + 4u * (lane & 3u) + ((lane >> 2) & 3u));
  auto request_sections = [&](TrailChunk<L>& c) {   // reads clamped into the buffer's 16 bytes of slack
    c.start_live = uni(c.pf_start); c.start_live2 = uni(c.pf_start2);
    const uint32_t last = (uint32_t)((c.src_len + 12) >> 2);
    const uint32_t d0 = (c.start_live >> 5) + lane, d1 = d0 + 64, e0 = (c.start_live2 >> 5) + lane;
    c.sec = load_u32_le(c.src + 4ull * (d0 < last ? d0 : last));
    c.sec_hi = load_u32_le(c.src + 4ull * (d1 < last ? d1 : last));
    c.sec_b = load_u32_le(c.src + 4ull * (e0 < last ? e0 : last));
  };
  uint32_t pv_next = 0;
  // One chunk after the other (the registers of four unpackings at once do not fit beside the walker: what overlaps within a chunk is its
  // two variables, across chunks the SIMD's other expander waves)
#pragma unroll
  for (int q = 0; q < 2; q++) {
    TrailChunk<L>& c = S[q];
    if (!c.sec_valid()) request_sections(c);
    // ---- stage A of the primary variable: symbols -> offset bits -> their prefix ----
    uint32_t syms, obs, excl;
    {
      const uint32_t mine = bperm(layout, c.pf_syms);
      syms = c.n_bins() <= 1 ? 0u : quad_transpose_u8(mine, lane & 3);
      uint32_t t = 0, o4 = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) { const uint32_t o = bperm(4u * ((syms >> (8 * k)) & 63u), c.tbl_ob); o4 |= o << (8 * k); t += o; }
      obs = o4; excl = wave_incl_scan(t) - t;
    }
    const uint32_t mine2 = bperm(layout, c.pf_syms2);   // (before pf_syms2 is requested anew)
    // ---- behind the section loads: the next batch's requests, the next iteration's progress words ----
    c.set(kFlHave, false);
    if (c.next + 1 < c.n_batches && ready[q] > c.next + 1) trail_request<L, true>(c, c.next + 1, ar);
    if (q == 1) pv_next = ld_agent(pline);
    // ---- stage B ----
    L x[4], y2[4];
#pragma unroll
    for (int k = 0; k < 4; k++) x[k] = bperm_t<L>(4u * ((syms >> (8 * k)) & 63u), c.tbl_low);
    if (c.max_ob() != 0) {
      const uint32_t rel = (c.start_live & 31u) + excl, di = rel >> 5, sh = rel & 31u;
      auto fetch = [&](uint32_t i) -> uint32_t { const uint32_t a = bperm(4u * (i & 63u), c.sec), b = bperm(4u * (i & 63u), c.sec_hi); return i < 64 ? a : b; };
      const uint32_t w0 = fetch(di), w1 = fetch(di + 1), w2 = fetch(di + 2);
      uint64_t v64 = ((uint64_t)__builtin_amdgcn_alignbit(w2, w1, sh) << 32) | __builtin_amdgcn_alignbit(w1, w0, sh);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t o = (obs >> (8 * k)) & 0xffu;
        x[k] = (L)(x[k] + (L)__builtin_amdgcn_ubfe((uint32_t)v64, 0u, o));
        v64 >>= o;
      }
    }
    if (c.dord()) trail_delta<L>(x, c.dord(), c.mom);
    {
      // the secondary variable, start to end (its symbols and section were requested with the primary's)
      const uint32_t syms2 = c.n_bins2() <= 1 ? 0u : quad_transpose_u8(mine2, lane & 3);
#pragma unroll
      for (int k = 0; k < 4; k++) y2[k] = bperm_t<L>(4u * ((syms2 >> (8 * k)) & 63u), c.tbl_low2);
      if (c.max_ob2() != 0) {   // (uniform)
        uint32_t t = 0, o4 = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { const uint32_t o = bperm(4u * ((syms2 >> (8 * k)) & 63u), c.tbl_ob2); o4 |= o << (8 * k); t += o; }
        const uint32_t rel = (c.start_live2 & 31u) + (wave_incl_scan(t) - t), di = rel >> 5, sh = rel & 31u;
        const uint32_t w0 = bperm(4u * di, c.sec_b), w1 = bperm(4u * di + 4u, c.sec_b), w2 = bperm(4u * di + 8u, c.sec_b);
        uint64_t v64 = ((uint64_t)__builtin_amdgcn_alignbit(w2, w1, sh) << 32) | __builtin_amdgcn_alignbit(w1, w0, sh);
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const uint32_t o = (o4 >> (8 * k)) & 0xffu;
          y2[k] = (L)(y2[k] + (L)__builtin_amdgcn_ubfe((uint32_t)v64, 0u, o));
          v64 >>= o;
        }
      }
    }
    L outv[4];
    trail_join4<L>(c.mode_kind(), c.num_kind(), c.mode_base, c.mode_k, x, y2, outv);
    L PCO_GLOBAL* o = c.dst + (uint64_t)c.next * kBatchN + 4 * lane;
    c.set(kFlSecValid, false);
    if (c.have() && c.n - (c.next + 1) * kBatchN >= kBatchN + c.nlps()) { request_sections(c); c.set(kFlSecValid, true); }   // (a full batch follows and its starts have been asked for; before the stores: trail_fast_pair)
    if constexpr (sizeof(L) == 8) {
      const unsigned long long y[4] = {outv[0], outv[1], outv[2], outv[3]};
      store_u64_batch((unsigned long long PCO_GLOBAL*)(c.dst + (uint64_t)c.next * kBatchN), y);
    } else if constexpr (sizeof(L) == 4) {
      trail_u32x4 a; a.x = outv[0]; a.y = outv[1]; a.z = outv[2]; a.w = outv[3];
      __builtin_nontemporal_store(a, (trail_u32x4 PCO_GLOBAL*)o);
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) o[k] = outv[k];
    }
    c.next++;
    __builtin_amdgcn_sched_barrier(0);   // (chunk after chunk: see above)
  }
  return pv_next;
}

// trail_fast_pair2 for chunks whose secondary variable is a CONSTANT (one bin, no offset bits: the adjustments of exact decimals under float-mult,
// BASELINE configs[2]) -- round 6.  Such a batch is ONE unpacking and a join, so the wave's two chunks go through it interleaved like
// trail_fast_pair's (trail_fast_pair2 takes them one after the other: four unpackings do not fit the registers, and its expanders finished
// 2.6 ms behind the walker), the primary's section in two registers (offsets of up to 15 bits), nothing of the secondary fetched at all.
template <class L>
__device__ __forceinline__ uint32_t trail_fast_pair2t(TrailChunk<L> (&S)[kTrailSlotsPerWave], const uint32_t (&ready)[kTrailSlotsPerWave], const uint32_t* pline, const TrailAreas& ar) {
  static_assert(kTrailSlotsPerWave == 2, "written for two chunks per wave");
  const uint32_t lane = lane_id();
  const uint32_t layout = 4u * ((lane & 48u) + 4u * (lane & 3u) + ((lane >> 2) & 3u));
  uint32_t syms[2], obs[2], excl[2];
  auto request_section = [&](TrailChunk<L>& c) {   // two registers: at most 120 + 3 dwords; reads clamped into the buffer's 16 bytes of slack
    c.start_live = uni(c.pf_start);
    const uint32_t last = (uint32_t)((c.src_len + 12) >> 2);
    const uint32_t d0 = (c.start_live >> 5) + lane, d1 = d0 + 64;
    c.sec = load_u32_le(c.src + 4ull * (d0 < last ? d0 : last));
    c.sec_hi = load_u32_le(c.src + 4ull * (d1 < last ? d1 : last));
  };
#pragma unroll
  for (int q = 0; q < 2; q++) {
    TrailChunk<L>& c = S[q];
    if (!c.sec_valid()) request_section(c);
    syms[q] = quad_transpose_u8(bperm(layout, c.pf_syms), lane & 3);
    uint32_t t = 0, o4 = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { const uint32_t o = bperm(4u * ((syms[q] >> (8 * k)) & 63u), c.tbl_ob); o4 |= o << (8 * k); t += o; }
    obs[q] = o4; excl[q] = wave_incl_scan(t) - t;
  }
#pragma unroll
  for (int q = 0; q < 2; q++) {
    TrailChunk<L>& c = S[q];
    c.set(kFlHave, false);
    if (c.next + 1 < c.n_batches && ready[q] > c.next + 1) trail_request<L, true>(c, c.next + 1, ar);
  }
  const uint32_t pv_next = ld_agent(pline);
#pragma unroll
  for (int q = 0; q < 2; q++) {
    TrailChunk<L>& c = S[q];
    L x[4], y2[4], outv[4];
#pragma unroll
    for (int k = 0; k < 4; k++) x[k] = bperm_t<L>(4u * ((syms[q] >> (8 * k)) & 63u), c.tbl_low);
    const uint32_t rel = (c.start_live & 31u) + excl[q], di = rel >> 5, sh = rel & 31u;
    auto fetch = [&](uint32_t i) -> uint32_t { const uint32_t a = bperm(4u * (i & 63u), c.sec), b = bperm(4u * (i & 63u), c.sec_hi); return i < 64 ? a : b; };
    const uint32_t w0 = fetch(di), w1 = fetch(di + 1), w2 = fetch(di + 2);
    uint64_t v64 = ((uint64_t)__builtin_amdgcn_alignbit(w2, w1, sh) << 32) | __builtin_amdgcn_alignbit(w1, w0, sh);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t o = (obs[q] >> (8 * k)) & 0xffu;
      x[k] = (L)(x[k] + (L)__builtin_amdgcn_ubfe((uint32_t)v64, 0u, o));
      v64 >>= o;
    }
    if (c.dord()) trail_delta<L>(x, c.dord(), c.mom);
    const L y2c = bperm_t<L>(0u, c.tbl_low2);   // bin 0's lower: the secondary latent of every number (lane 0 holds it; zero without a bin)
#pragma unroll
    for (int k = 0; k < 4; k++) y2[k] = y2c;
    trail_join4<L>(c.mode_kind(), c.num_kind(), c.mode_base, c.mode_k, x, y2, outv);
    L PCO_GLOBAL* o = c.dst + (uint64_t)c.next * kBatchN + 4 * lane;
    c.set(kFlSecValid, false);
    if (c.have() && c.n - (c.next + 1) * kBatchN >= kBatchN + c.nlps()) { request_section(c); c.set(kFlSecValid, true); }   // (a full batch follows and its start has been asked for; before the stores: trail_fast_pair)
    if constexpr (sizeof(L) == 8) {
      const unsigned long long y[4] = {outv[0], outv[1], outv[2], outv[3]};
      store_u64_batch((unsigned long long PCO_GLOBAL*)(c.dst + (uint64_t)c.next * kBatchN), y);
    } else if constexpr (sizeof(L) == 4) {
      trail_u32x4 a; a.x = outv[0]; a.y = outv[1]; a.z = outv[2]; a.w = outv[3];
      __builtin_nontemporal_store(a, (trail_u32x4 PCO_GLOBAL*)o);
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) o[k] = outv[k];
    }
    c.next++;
  }
  return pv_next;
}

#ifdef PCO_TRAIL_TIMING
__device__ unsigned long long g_trail_timing[8];   // block 0, wave 0: iterations, poll, stage A, requests, stage B (s_memtime units), idle polls
#define TT_NOW() __builtin_readcyclecounter()
#define TT_ADD(i, t0) do { const unsigned long long _n = TT_NOW(); tt[i] += _n - (t0); (t0) = _n; } while (0)
#else
#define TT_ADD(i, t0) do { } while (0)
#endif
// kTwo: the expanders of the walker blocks that hold a chunk with TWO latent variables (int-mult / float-mult / float-quant; decided per block
// by the same glance at the chunk preambles the walkers use, block_trail_kinds): such a chunk's batch is two unpackings and a join, so its
// state is twice a classic chunk's -- a kernel of its own keeps that out of the registers of the common one.  Every walker block is followed
// by exactly one of the two kernels; the other's block for it leaves at once.
template <class L, bool kTwo>
__global__ __launch_bounds__(64 * kTrailWaves) __attribute__((amdgpu_waves_per_eu(5, 5))) void dec_trail_kernel(const PcoGfxDecodeTask* tasks, const uint32_t* task_ids, uint32_t n_ids, DecPlan* plans,
                                                                     const uint8_t* bins_area, const uint8_t* sym_area, uint64_t sym_stride,
                                                                     const uint64_t* offpos_area, uint64_t offpos_stride, const uint32_t* progress,
                                                                     uint32_t n_walk_blocks, const MetaRef* metas) {
  const uint32_t lane = lane_id(), wave = uni(threadIdx.x >> 6);
  const TrailAreas ar{sym_area, sym_stride, offpos_area, offpos_stride};
  for (uint32_t wb = blockIdx.x; wb < n_walk_blocks; wb += gridDim.x) {
    if (((block_trail_kinds(tasks, task_ids, n_ids, wb, metas) & 2u) != 0) != kTwo) continue;   // the other kernel's block
    if (wave == 0) PCO_TRAIL_STAMP(2, wb);
    TrailChunk<L> S[kTrailSlotsPerWave];
    const uint32_t* pline = progress + (uint64_t)wb * kTrailProgressStride + (lane & 7u);
    // ---- the chunks' constants, once their walker has parsed the metadata (it publishes all eight slots together) ----
    uint32_t pv = 0;
    {
      uint32_t tries = 0;
      for (;;) {
        pv = ld_agent(pline);
        bool all = true;
#pragma unroll
        for (int q = 0; q < (int)kTrailSlotsPerWave; q++) if (wb * 8 + wave * kTrailSlotsPerWave + q < n_ids && (uint32_t)__builtin_amdgcn_readlane((int)pv, (int)(wave * kTrailSlotsPerWave + q)) == 0) all = false;
        if (all || ++tries > kTrailSpinLimit) break;
        __builtin_amdgcn_s_sleep(64);
      }
    }
#pragma unroll
    for (int q = 0; q < (int)kTrailSlotsPerWave; q++) {
      TrailChunk<L>& c = S[q];
      const uint32_t slot = wave * kTrailSlotsPerWave + q, bi = wb * 8 + slot;
      c.fl = bi < n_ids ? kFlLive : 0u; c.next = 0;
      const uint32_t p0 = (uint32_t)__builtin_amdgcn_readlane((int)pv, (int)slot);
      if (p0 == 0 || p0 == kTrailDead) c.fl = 0;   // (0: the walker did not start within the time limit -- it may still mark the chunk fused later; without this wave's done mark dec_expand_kernel takes it)
      c.ti = c.live() ? (task_ids ? uni(task_ids[bi]) : bi) : 0u;
      // the plan is 64 words: lane i fetches word i (one request), the fields come out of the lanes
      const uint32_t pw = ld_agent((const uint32_t*)(plans + c.ti) + lane);
      auto word = [&](uint32_t i) -> uint32_t { return (uint32_t)__builtin_amdgcn_readlane((int)pw, (int)i); };
      auto word64 = [&](uint32_t i) -> uint64_t { return (uint64_t)word(i) | ((uint64_t)word(i + 1) << 32); };
      const uint32_t fused = word(kPlanFused);
      const bool two = kTwo && fused == 2;
      bool ok = c.live() && (two || fused == 1);
      // (the walker never marks a chunk this kernel cannot take; belt and braces)
      const uint32_t nb1 = word(kPlanNBins + 1), mo1 = word(kPlanMaxOb + 1), nb2 = two ? word(kPlanNBins + 2) : 0u, mo2 = two ? word(kPlanMaxOb + 2) : 0u;
      const uint32_t dord = word(kPlanDeltaKind + 1) == kDeltaConsecutive ? word(kPlanDeltaOrder + 1) : 0u;
      if (word(kPlanPresent) != 0 || nb1 > kTrailMaxBins || mo1 > 16 || dord > 2) ok = false;
      if (!two && (word(kPlanModeKind) != kClassic || word(kPlanPresent + 2) != 0)) ok = false;
      if (two && (word(kPlanPresent + 2) == 0 || nb2 > kTrailMaxBins || mo2 > 16 || word(kPlanDeltaKind + 2) != kDeltaNone)) ok = false;
      const PcoGfxDecodeTask* task = tasks + c.ti;
      c.n = ok ? word(kPlanN) : 0u; c.n_batches = (c.n + kBatchN - 1) / kBatchN;
      c.src = (gcptr_u8)(uintptr_t)uni((uint64_t)(uintptr_t)task->src); c.src_len = uni((uint64_t)task->src_len); c.dst = (L PCO_GLOBAL*)(uintptr_t)uni((uint64_t)(uintptr_t)task->dst);
      c.shape = uni((word(kPlanNumKind) & 3u) | ((nb1 & 127u) << 2) | ((mo1 & 31u) << 9) | ((dord & 3u) << 14) | ((word(kPlanModeKind) & 7u) << 16) | ((nb2 & 127u) << 19) | ((mo2 & 31u) << 26));
      c.mom[0] = (L)word64(kPlanMoments); c.mom[1] = (L)word64(kPlanMoments + 2);
      c.tbl_low = 0; c.tbl_ob = 0; c.pf_syms = 0; c.pf_start = 0; c.start_live = 0; c.sec = 0;
      const bool aligned = (((uintptr_t)c.dst) & 15) == 0;
      uint32_t fl = ok ? kFlLive : 0u;
      if (ok && two) fl |= kFlTwo;
      if (ok && !two && nb1 > 1 && mo1 >= 1 && mo1 <= 7 && aligned) fl |= kFlFast;
      if (ok && two && mo1 <= 15 && mo2 <= 7 && aligned) fl |= kFlFast2;
      if (ok && two && nb1 > 1 && mo1 >= 1 && mo1 <= 15 && nb2 <= 1 && mo2 == 0 && aligned) fl |= kFlTriv2;
      c.fl = uni(fl);
      if constexpr (kTwo) {
        c.mode_k = word(kPlanModeK); c.mode_base = (L)word64(kPlanModeBase);
        c.tbl_low2 = 0; c.tbl_ob2 = 0; c.pf_syms2 = 0; c.pf_start2 = 0; c.start_live2 = 0; c.sec_hi = 0; c.sec_b = 0;
        if (c.two() && lane < nb2) {
          const uint8_t* bins2 = bins_area + (uint64_t)c.ti * kBinsAreaPerTask + 2 * kBinsAreaPerVar;   // the secondary variable's area
          c.tbl_low2 = (L)ld_agent((const uint64_t*)bins2 + lane);
          c.tbl_ob2 = (ld_agent((const uint32_t*)(bins2 + kFastMaxBins * 8) + (lane >> 2)) >> (8 * (lane & 3))) & 0xffu;
        }
      }
      if (c.live() && lane < nb1) {
        const uint8_t* bins = bins_area + (uint64_t)c.ti * kBinsAreaPerTask + kBinsAreaPerVar;   // the primary variable's area
        c.tbl_low = (L)ld_agent((const uint64_t*)bins + lane);
        c.tbl_ob = (ld_agent((const uint32_t*)(bins + kFastMaxBins * 8) + (lane >> 2)) >> (8 * (lane & 3))) & 0xffu;
      }
    }
    // ---- batch after batch, the wave's chunks in lockstep ----
    uint32_t idle = 0;
#ifdef PCO_TRAIL_TIMING
    unsigned long long tt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tt0 = TT_NOW();
#endif
    for (;;) {
      bool any_live = false, did = false;
      uint32_t ready[kTrailSlotsPerWave];
#pragma unroll
      for (int q = 0; q < (int)kTrailSlotsPerWave; q++) {
        TrailChunk<L>& c = S[q];
        ready[q] = 0;
        if (!c.live()) continue;
        if (c.next >= c.n_batches) {   // expanded to the end: the mark dec_expand_kernel looks for
          if (lane == 0) __hip_atomic_store((uint32_t*)progress + (uint64_t)wb * kTrailProgressStride + kTrailDoneWord + wave * kTrailSlotsPerWave + q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          c.set(kFlLive, false); continue;
        }
        any_live = true;
        const uint32_t p = (uint32_t)__builtin_amdgcn_readlane((int)pv, (int)(wave * kTrailSlotsPerWave + q));
        if (p == kTrailDead) { c.set(kFlLive, false); continue; }   // the walker met an error: it reports it, nothing more to expand
#ifdef PCO_TRAIL_CHEAT   // (measurement builds only: take the previous, identical call's symbols as if the walker had finished -- contention without the hand-over)
        const uint32_t done = c.n_batches;
#else
        const uint32_t done = p - 1;                          // batches whose symbols and section starts are out
#endif
        if (done <= c.next) continue;
        if (!c.have()) trail_request<L, kTwo>(c, c.next, ar);
        // stay one batch behind the walker unless it has finished the chunk: the requests of the batch after this one can then go out
        // under this one's windows, and have the rest of the iteration to come back
        if (done < c.n_batches && done < c.next + 2) continue;
        ready[q] = done;
      }
      if (!any_live) break;
      TT_ADD(1, tt0);
#ifndef PCO_TRAIL_NOFAST
      if (ready[0] && ready[1] && S[0].fast_ok() && S[1].fast_ok() && S[0].n - S[0].next * kBatchN >= kBatchN + S[0].nlps() && S[1].n - S[1].next * kBatchN >= kBatchN + S[1].nlps()) {
        pv = trail_fast_pair<L>(S, ready, pline, ar);
        idle = 0;
        TT_ADD(4, tt0);
#ifdef PCO_TRAIL_TIMING
        tt[0]++; tt[6]++;
#endif
        continue;
      }
      if constexpr (kTwo) {
        if (ready[0] && ready[1] && S[0].triv2_ok() && S[1].triv2_ok() && S[0].n - S[0].next * kBatchN >= kBatchN + S[0].nlps() && S[1].n - S[1].next * kBatchN >= kBatchN + S[1].nlps()) {
          pv = trail_fast_pair2t<L>(S, ready, pline, ar);
          idle = 0;
          TT_ADD(4, tt0);
#ifdef PCO_TRAIL_TIMING
          tt[0]++; tt[6]++;
#endif
          continue;
        }
        if (ready[0] && ready[1] && S[0].fast2_ok() && S[1].fast2_ok() && S[0].n - S[0].next * kBatchN >= kBatchN + S[0].nlps() && S[1].n - S[1].next * kBatchN >= kBatchN + S[1].nlps()) {
          pv = trail_fast_pair2<L>(S, ready, pline, ar);
          idle = 0;
          TT_ADD(4, tt0);
#ifdef PCO_TRAIL_TIMING
          tt[0]++; tt[6]++;
#endif
          continue;
        }
      }
#endif
      uint32_t pv_next = 0;
      if constexpr (!kTwo) {
        TrailItem it[kTrailSlotsPerWave];
        uint32_t cnts[kTrailSlotsPerWave];
  #pragma unroll
        for (int q = 0; q < (int)kTrailSlotsPerWave; q++) it[q].s1 = it[q].s2 = 0;   // (before anything of this iteration is in flight)
  #pragma unroll
        for (int q = 0; q < (int)kTrailSlotsPerWave; q++) {
          TrailChunk<L>& c = S[q];
          if (!ready[q]) continue;
          did = true;
          cnts[q] = trail_cnt(c, c.next);
          c.start_live = uni(c.pf_start);
          trail_section(trail_primary(c), c.src, c.src_len, cnts[q], it[q]);      // (first: it only waits for the section start; the symbol work below runs under it)
          trail_stage_a(trail_primary(c), cnts[q], it[q]);
        }
        TT_ADD(2, tt0);
        // behind the windows (loads return in order: a request issued before them would be waited for with them): the requests for the
        // batches after these, and the progress words for the next iteration -- trips to another XCD's memory side, microseconds long
  #pragma unroll
        for (int q = 0; q < (int)kTrailSlotsPerWave; q++) {
          TrailChunk<L>& c = S[q];
          if (!ready[q]) continue;
          c.set(kFlHave, false);
          if (c.next + 1 < c.n_batches && ready[q] > c.next + 1) trail_request<L, false>(c, c.next + 1, ar);
        }
        pv_next = ld_agent(pline);
        TT_ADD(3, tt0);
  #pragma unroll
        for (int q = 0; q < (int)kTrailSlotsPerWave; q++) {
          TrailChunk<L>& c = S[q];
          if (!ready[q]) continue;
          const uint32_t b = c.next;
          L prim[4];
          trail_stage_b(trail_primary(c), cnts[q], it[q], prim);
          if (c.dord()) trail_delta<L>(prim, c.dord(), c.mom);
          L outv[4];
  #pragma unroll
          for (int k = 0; k < 4; k++) outv[k] = from_latent_ordered<L>(prim[k], c.num_kind());   // classic join (mode/classic.rs:14-24)
          const uint32_t j0 = b * kBatchN, n_remaining = c.n - j0, batch_n = n_remaining < kBatchN ? n_remaining : kBatchN;
          const uint32_t i0 = 4 * lane;
          L PCO_GLOBAL* o = c.dst + j0 + i0;
          if (i0 + 4 <= batch_n && (((uintptr_t)o) & 15) == 0) {
            if constexpr (sizeof(L) == 8) {
              typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
              u64x2 PCO_GLOBAL* p2 = (u64x2 PCO_GLOBAL*)o;
              u64x2 a; a.x = outv[0]; a.y = outv[1]; u64x2 bb; bb.x = outv[2]; bb.y = outv[3];
              p2[0] = a; p2[1] = bb;
            } else if constexpr (sizeof(L) == 4) {
              trail_u32x4 a; a.x = outv[0]; a.y = outv[1]; a.z = outv[2]; a.w = outv[3];
              *(trail_u32x4 PCO_GLOBAL*)o = a;
            } else { for (int k = 0; k < 4; k++) o[k] = outv[k]; }
          } else { for (int k = 0; k < 4; k++) if (i0 + k < batch_n) o[k] = outv[k]; }
          c.next = b + 1; c.set(kFlSecValid, false);
        }
      } else {
        // one chunk after the other, each from its requests to its stores (the registers of two chunks of two variables in lockstep do not fit)
#pragma unroll
        for (int q = 0; q < (int)kTrailSlotsPerWave; q++) {
          TrailChunk<L>& c = S[q];
          if (!ready[q]) continue;
          did = true;
          TrailItem it, it2;
          it.s1 = it.s2 = 0; it2.s0 = it2.s1 = it2.s2 = 0; it2.syms = it2.obs = it2.excl = 0;
          const uint32_t b = c.next, cnt = trail_cnt(c, b);
          const uint32_t left = c.n - b * kBatchN, cnt2 = left < kBatchN ? left : kBatchN;   // the secondary variable: no delta, a batch holds min(256, what is left of the chunk) of its latents
          c.start_live = uni(c.pf_start);
          trail_section(trail_primary(c), c.src, c.src_len, cnt, it);
          if (c.two()) { c.start_live2 = uni(c.pf_start2); trail_section(trail_secondary(c), c.src, c.src_len, cnt2, it2); }
          trail_stage_a(trail_primary(c), cnt, it);
          if (c.two()) trail_stage_a(trail_secondary(c), cnt2, it2);
          c.set(kFlHave, false);
          if (b + 1 < c.n_batches && ready[q] > b + 1) trail_request<L, true>(c, b + 1, ar);
          L prim[4], outv[4];
          trail_stage_b(trail_primary(c), cnt, it, prim);
          if (c.dord()) trail_delta<L>(prim, c.dord(), c.mom);
          if (c.two()) {   // mode/{int_mult,float_mult,float_quant}.rs join_latents, as dec_expand_kernel
            L sec[4];
            trail_stage_b(trail_secondary(c), cnt2, it2, sec);
            trail_join4<L>(c.mode_kind(), c.num_kind(), c.mode_base, c.mode_k, prim, sec, outv);
          } else {
#pragma unroll
            for (int k = 0; k < 4; k++) outv[k] = from_latent_ordered<L>(prim[k], c.num_kind());   // classic join (mode/classic.rs:14-24)
          }
          const uint32_t j0 = b * kBatchN, batch_n = left < kBatchN ? left : kBatchN;
          const uint32_t i0 = 4 * lane;
          L PCO_GLOBAL* o = c.dst + j0 + i0;
          if (i0 + 4 <= batch_n && (((uintptr_t)o) & 15) == 0) {
            if constexpr (sizeof(L) == 8) {
              typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
              u64x2 PCO_GLOBAL* p2 = (u64x2 PCO_GLOBAL*)o;
              u64x2 a; a.x = outv[0]; a.y = outv[1]; u64x2 bb; bb.x = outv[2]; bb.y = outv[3];
              p2[0] = a; p2[1] = bb;
            } else if constexpr (sizeof(L) == 4) {
              trail_u32x4 a; a.x = outv[0]; a.y = outv[1]; a.z = outv[2]; a.w = outv[3];
              *(trail_u32x4 PCO_GLOBAL*)o = a;
            } else { for (int k = 0; k < 4; k++) o[k] = outv[k]; }
          } else { for (int k = 0; k < 4; k++) if (i0 + k < batch_n) o[k] = outv[k]; }
          c.next = b + 1; c.set(kFlSecValid, false);
          __builtin_amdgcn_sched_barrier(0);
        }
        pv_next = ld_agent(pline);
      }
      TT_ADD(4, tt0);
#ifdef PCO_TRAIL_TIMING
      tt[0]++; if (!did) tt[5]++;
#endif
      if (did) { idle = 0; pv = pv_next; continue; }
      __builtin_amdgcn_s_sleep(127);
      pv = ld_agent(pline);
      if (++idle > kTrailSpinLimit) {
        // no progress for ~55 ms: the walker is not running beside us (or is stuck).  Leave: the chunks this wave has not marked done are
        // expanded by dec_expand_kernel from their first batch, after both kernels have ended.
        break;
      }
    }
#ifdef PCO_TRAIL_TIMING
    if (blockIdx.x == 0 && wave == 0 && lane == 0) for (int i = 0; i < 8; i++) g_trail_timing[i] = tt[i];
    __syncthreads();   // (the block's last wave)
    if (wave == 0) PCO_TRAIL_STAMP(3, wb);
#endif
  }
}

}  // namespace pcogfx
