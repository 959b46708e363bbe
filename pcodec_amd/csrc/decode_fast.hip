// decode_fast.hip -- two-kernel chunk decoder for the common case (one chunk per task, tANS tables that fit
// a 5.5 KB LDS slice).  The single-kernel decoder in decode_kernel.hip stays as the general fallback.
//
// Why two kernels: profiling the one-wave-per-chunk kernel showed it is ISSUE-bound in the tANS walk -- the
// four interleaved chains of a chunk occupy 4 of 64 lanes, so 64 steps x ~50 instructions per batch are paid
// per chunk.  Here
//   dec_walk_kernel   packs EIGHT chunks into one wave (8 groups of 8 lanes; lanes 0-3 of a group walk that
//                     chunk's 4 tANS chains in lock-step with the other groups), parses all metadata, builds the
//                     decoder tables in LDS and emits one u16 bin symbol per latent plus the bit position of
//                     every batch's offset section (page_latent_decompressor.rs:89-177);
//   dec_expand_kernel is the embarrassingly parallel rest, one wave per chunk: offsets unpack, delta scans,
//                     join, coalesced stores (page_latent_decompressor.rs:15-44, delta/*.rs, mode/*.rs).
// Extra HBM traffic versus the fused kernel: 2 B written + 2 B read per latent.
#include "decode_kernel.hip"

namespace pcogfx {

// dec_walk_kernel<L, KQ>: KQ chunks per wave -- 8 (tables up to 4.2 KB: one variable at ans_size_log 10) or 4 (9.1 KB) -- FOUR
// LANES (one per tANS chain) per chunk.  Tasks whose tables do not fit the 8-chunk slices are handed to the 4-chunk stage
// through DecPlan::status.  (Packing 16 per wave was tried: level-8 chunks mostly need ans_size_log 9-10, whose tables do not
// fit a 2.4 KB slice, and a launch that splits its chunks over two stages pays the walk latency twice.)  Slice of a chunk: u64[56] ANS window | VarInfo[3] | per variable: entries u32[T], offset bits of bin 0 (8 B).
constexpr uint32_t kGrpWinOff = 0;
constexpr uint32_t kGrpVarOff = 448;
constexpr uint32_t kGrpTblOff = 640;
template <uint32_t KQ> struct WalkCfg {
  static constexpr uint32_t kGrpBytes = KQ == 8 ? 4912u : 9904u;   // (x 4 + the 1312 B of scratch = 40928 B: four blocks per CU in either form)
  static constexpr uint32_t kGrpTblBytes = kGrpBytes - kGrpTblOff;      // 4272 / 9264
  static constexpr uint32_t kWalkTmpOff = KQ * kGrpBytes;                // scratch for the table build (one chunk at a time): u32[264] cumulative weights | u8[256] offset bits
  static constexpr uint32_t kWalkTmpObOff = kWalkTmpOff + 1056;
  static constexpr uint32_t kWalkLdsBytes = kWalkTmpObOff + 256;         // 40608 / 40928: four waves per CU
  static constexpr uint32_t kRetryStatus = KQ == 8 ? 101u : 100u;       // where a task goes whose tables do not fit
  static_assert(kGrpBytes % 16 == 0, "chunk slices must stay 16-byte aligned");
  static_assert(kWalkLdsBytes < 65536, "walk entries hold 16-bit LDS addresses");
};
// A walker "block" is one wave and its LDS.  The publishing walker's workgroups hold FOUR of them (256 threads, 4 x 40608 B of LDS: the whole
// CU), because a four-wave workgroup is dealt one wave to each SIMD, while four one-wave workgroups land wherever the dispatcher's pointer
// stands -- with a second kernel's blocks arriving in between, two walkers on one SIMD and none on another was common: the two slowed each
// other (8.4 ms instead of 7.5), and their 256 registers left that SIMD room for two expander waves instead of four, so one expander block
// in twenty started when the walk was over and expanded its eight chunks alone (scripts/trail_timing.py: the 2.6 ms tail of round 4's first form).
template <uint32_t KQ> __device__ __forceinline__ uint8_t PCO_LDS* walk_lds() { return lds_base() + uni((uint32_t)(threadIdx.x >> 6)) * WalkCfg<KQ>::kWalkLdsBytes; }
__device__ __forceinline__ uint32_t walk_block_id() { return blockIdx.x * (blockDim.x >> 6) + uni((uint32_t)(threadIdx.x >> 6)); }
// Trailing expanders (decode_trail.hip): dec_walk_kernel<L, 8, true> publishes, per chunk slot, how many batches of the chunk are complete
// in global memory -- progress[block * 8 + slot] = 1 + batches once the metadata is parsed, kTrailDead for a chunk the expanders must
// leave alone -- and a second kernel on a second stream expands them while the walk is still going on.
constexpr uint32_t kTrailDead = 0xffffffffu;
constexpr uint32_t kTrailDoneWord = 8;          // ... and words 8..15 of that line: slot's chunk has been expanded to the end by its expander wave (decode_trail.hip)
constexpr uint32_t kTrailProgressStride = 32;   // words per walker block: its eight progress words have a 128-byte line to themselves (the expanders of other blocks poll theirs)
#ifdef PCO_TRAIL_NODEFER
constexpr bool kTrailDefer = false;
#else
constexpr bool kTrailDefer = true;    // the walker's agent-scope stores go out one round late (see dec_walk_body)
#endif
constexpr uint32_t kTrailMaxBins = 64;        // a variable's bins live in the registers of one wave, a bin per lane
constexpr uint32_t kFastMaxBins = 256;
constexpr uint32_t kStatusRetryLegacy = 100;     // internal: hand the task to the single-kernel decoder
constexpr uint32_t kStatusRetryK4 = 101;         // internal: tables too big for an 8-chunk wave, try the 4-chunk walker
constexpr uint32_t kStatusNeedHist = 102;        // internal: lookback with a delta'd secondary variable -- the task comes again with scratch for that history

struct DecPlan {   // written by dec_walk_kernel, read by dec_expand_kernel
  uint32_t status, n;
  uint32_t mode_kind, mode_k;
  uint64_t mode_base;
  uint32_t num_kind, dtype;
  uint32_t present[3], n_bins[3], max_ob[3], delta_kind[3], delta_order[3], nlps[3];
  uint32_t window_n_log, state_n_log;
  uint64_t moments[2][8];
  uint64_t consumed;
  uint32_t fused, more;     // fused = 1: the chunk is expanded by dec_trail_kernel while the walk runs (its result is written by the walker), dec_expand_kernel skips it; more: PCO_GFX_TASK_ONE_CHUNK, bit 0 = another chunk follows, bit 1 = the terminator was consumed
};
constexpr uint64_t kBinsAreaPerVar = kFastMaxBins * 8 + kFastMaxBins;   // lowers (8 B stride) then offset bits
constexpr uint64_t kBinsAreaPerTask = 3 * kBinsAreaPerVar;

// walk entry (one u32 per tANS state): bits 0..5 bits_to_read (<= 11; the two spare bits keep the sum of three of them
// inside the field), 6..12 the bin's offset bits, 13..20 the bin symbol, 21..31 next_state_idx_base - T (the walker adds
// the bits it read and turns the sum into an LDS address with the table's base).
__device__ __forceinline__ uint32_t make_wentry(uint32_t next_base, uint32_t sym, uint32_t btr, uint32_t ob) { return btr | (ob << 6) | (sym << 13) | (next_base << 21); }
// (a table with ans_size_log 12 does not fit a slice, so 11 bits of state index are enough)

// Build one variable's walk table for chunk slot `q`: u32 entries and per-bin offset bits in LDS; lowers / offset bits
// go to the global bins area for dec_expand_kernel.  All 64 lanes cooperate.  (ans/spec.rs:37-59, ans/decoding.rs:27-47)
template <class LV, uint32_t KQ>
__device__ __forceinline__ bool fast_build_var_impl(uint32_t q, uint32_t vi, MetaReader& mr, uint8_t PCO_GLOBAL* bins_out, uint32_t& status) {
  constexpr uint32_t kGrpBytes = WalkCfg<KQ>::kGrpBytes, kWalkTmpOff = WalkCfg<KQ>::kWalkTmpOff;
  const uint32_t lane = lane_id();
  uint8_t PCO_LDS* grp = walk_lds<KQ>() + q * kGrpBytes;
  VarInfo PCO_LDS* vinfo = (VarInfo PCO_LDS*)(grp + kGrpVarOff) + vi;
  const uint32_t asl = uni(vinfo->ans_size_log), n_bins = uni(vinfo->n_bins), latent_bits = uni(vinfo->latent_bits);
  const uint32_t T = 1u << asl;
  const uint32_t tbl_addr = (uint32_t)(uintptr_t)walk_lds<KQ>() + q * kGrpBytes + kGrpTblOff + uni(vinfo->off_nodes);   // absolute LDS byte address
  uint32_t PCO_LDS* entries = (uint32_t PCO_LDS*)(grp + kGrpTblOff + uni(vinfo->off_nodes));
  uint8_t PCO_LDS* obs = walk_lds<KQ>() + WalkCfg<KQ>::kWalkTmpObOff;   // every bin's offset bits, for the entries being built
  uint8_t PCO_LDS* ob0 = grp + kGrpTblOff + uni(vinfo->off_ob);      // the slice keeps bin 0's only (what a single-bin variable needs)
  uint32_t PCO_LDS* cum = (uint32_t PCO_LDS*)(walk_lds<KQ>() + kWalkTmpOff);
  uint64_t PCO_GLOBAL* g_low = (uint64_t PCO_GLOBAL*)(bins_out + (uint64_t)vi * kBinsAreaPerVar);
  uint8_t PCO_GLOBAL* g_ob = bins_out + (uint64_t)vi * kBinsAreaPerVar + kFastMaxBins * 8;
  const uint32_t obb = offset_bits_bits(latent_bits);
  const uint32_t bin_bits = asl + latent_bits + obb;
  const uint64_t bins_start = mr.bit;
  uint32_t bad = 0, max_ob = 0, carry = 0;
  for (uint32_t b0 = 0; b0 < n_bins; b0 += 64) {
    const uint32_t b = b0 + lane;
    uint32_t w = 0;
    if (b < n_bins) {
      const uint64_t at = bins_start + (uint64_t)b * bin_bits;
      w = (uint32_t)mr.peek(at, asl) + 1;
      const uint64_t lower = mr.peek(at + asl, latent_bits);
      const uint32_t ob = (uint32_t)mr.peek(at + asl + latent_bits, obb);
      if (ob > latent_bits) bad = 1;
      g_low[b] = lower; g_ob[b] = (uint8_t)ob; obs[b] = (uint8_t)ob;
      if (b == 0) ob0[0] = (uint8_t)ob;
      max_ob = max_ob > ob ? max_ob : ob;
    }
    const uint32_t incl = wave_incl_scan(w);
    if (b < n_bins) cum[b] = carry + incl - w;
    carry += wave_last(incl);
  }
  if (lane == 0) cum[n_bins] = carry;
  mr.bit = bins_start + (uint64_t)n_bins * bin_bits;
  max_ob = uni(wave_max_u32(max_ob));
  if (lane == 0) vinfo->max_ob = max_ob;
  if (!mr.in_bounds()) { status = PCO_GFX_INSUFFICIENT_DATA; return false; }
  if (uni(wave_or_u32(bad))) { status = PCO_GFX_CORRUPTION; return false; }
  if (n_bins == 0) { if (lane == 0) entries[0] = make_wentry(0, 0, 0, 0); wave_sync_lds(); return true; }
  if (carry != T) { status = PCO_GFX_CORRUPTION; return false; }
  wave_sync_lds();
  uint32_t stride = (3 * T) / 5; if ((stride & 1) == 0) stride += 1;
  for (uint32_t t = lane; t < T; t += 64) {
    uint32_t lo = 0, hi = n_bins;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (cum[mid] <= t) lo = mid; else hi = mid; }
    entries[(stride * t) & (T - 1)] = lo;   // pass 1: the spread symbol only
  }
  wave_sync_lds();
  for (uint32_t b0 = 0; b0 < n_bins; b0 += 64) {
    const uint32_t b = b0 + lane;
    uint32_t w = 0;
    if (b < n_bins) w = cum[b + 1] - cum[b];
    wave_sync_lds();
    if (b < n_bins) cum[b] = w;
    wave_sync_lds();
  }
  const uint32_t sym_bits = 32 - clz_u32(n_bins - 1 > 0 ? n_bins - 1 : 1);
  for (uint32_t i0 = 0; i0 < T; i0 += 64) {
    const uint32_t i = i0 + lane;
    const bool act = i < T;
    const uint32_t s = act ? entries[i] : 0xffffffffu;
    uint64_t m = __ballot(act);
    for (uint32_t bit = 0; bit < sym_bits; bit++) { const uint64_t bm = __ballot((s >> bit) & 1); m &= ((s >> bit) & 1) ? bm : ~bm; }
    const uint64_t lt = ((uint64_t)1 << lane) - 1;
    const uint32_t rank = __popcll(m & lt), gcount = __popcll(m);
    const uint32_t basec = act ? cum[s] : 0;
    wave_sync_lds();
    if (act && rank == 0) cum[s] = basec + gcount;
    wave_sync_lds();
    if (act) { const uint32_t x_s = basec + rank; const uint32_t btr = clz_u32(x_s) - clz_u32(T); entries[i] = make_wentry((x_s << btr) - T, s, btr, obs[s]); }
  }
  wave_sync_lds();
  return true;
}

// (out of line for the ordinary walkers; the walker that shares its SIMD with the trailing expanders inlines it -- a called function
//  keeps its own register budget, and that kernel's is capped)
template <class LV, uint32_t KQ>
__device__ __noinline__ bool fast_build_var(uint32_t q, uint32_t vi, MetaReader& mr, uint8_t PCO_GLOBAL* bins_out, uint32_t& status) { return fast_build_var_impl<LV, KQ>(q, vi, mr, bins_out, status); }

struct FrontOut {
  uint32_t status, n, mode_kind;
  uint64_t bitpos;          // first bit of the page body
  uint32_t states[3][4];
  uint64_t moments[2][2];   // the first two delta moments of (primary, secondary)
};

// Everything before the page body for one task, executed by the whole wave on behalf of group q.
template <class L, uint32_t KQ, bool kInline>
__device__ __forceinline__ void fast_front_impl(const PcoGfxDecodeTask& task, uint32_t q, DecPlan PCO_GLOBAL* plan, uint8_t PCO_GLOBAL* bins_out, FrontOut& out,
                                                gcptr_u8 meta_p, uint64_t meta_len) {   // meta_p: a wrapped page whose ChunkMeta has a buffer of its own (MetaRef); else nullptr
  constexpr uint32_t kGrpBytes = WalkCfg<KQ>::kGrpBytes, kGrpTblBytes = WalkCfg<KQ>::kGrpTblBytes;
  const uint32_t lane = lane_id();
  gcptr_u8 src = (gcptr_u8)task.src;
  const uint64_t src_len = uni((uint64_t)task.src_len);
  const uint32_t dtype = uni(task.dtype), flags = uni(task.flags);
  const uint64_t dst_cap = uni((uint64_t)task.dst_cap);
  constexpr uint32_t LB = LBits<L>::v;
  const bool wrapped = (flags & PCO_GFX_TASK_WRAPPED_PAGE) != 0;
  MetaReader mr{src, src_len, 0};
  if (wrapped && meta_p != nullptr) mr = MetaReader{meta_p, meta_len, 0};
  uint32_t status = PCO_GFX_OK, format_major = 4, uniform_type = 0;
  out.status = PCO_GFX_OK; out.n = 0; out.bitpos = 0;
  for (int v = 0; v < 3; v++) for (int j = 0; j < 4; j++) out.states[v][j] = 0;
  out.mode_kind = kClassic;
  VarInfo PCO_LDS* vinfo = (VarInfo PCO_LDS*)(walk_lds<KQ>() + q * kGrpBytes + kGrpVarOff);
  // (a wrapped page that goes wrong -- here or in its body -- is handed to the single-kernel decoder, which also says how many batches came out
  //  before the failure: PageDecompressor::read's error timing, wrapped/page_decompressor.rs:193-221)
  auto fail = [&](uint32_t s) { if (wrapped && s != kStatusRetryK4) s = kStatusRetryLegacy; out.status = s; if (lane == 0) { plan->status = s; plan->consumed = mr.bit >> 3; plan->n = 0; plan->fused = 0; plan->more = 0; } };
  out.moments[0][0] = out.moments[0][1] = out.moments[1][0] = out.moments[1][1] = 0;
  if (dtype_bits(dtype) != (int)LB) { fail(PCO_GFX_INVALID_ARGUMENT); return; }
  if (flags & PCO_GFX_TASK_META_ONLY) { fail(kStatusRetryLegacy); return; }
  uint32_t n = 0;
  if (wrapped) {   // wrapped surface: ChunkMeta, then one page of exactly dst_cap numbers; the format version comes from the caller (wrapped/file_decompressor.rs:44-52)
    format_major = (flags >> 8) & 0xffu;
    if (dst_cap == 0 || dst_cap > kMaxEntries || format_major > 4) { fail(kStatusRetryLegacy); return; }
    n = (uint32_t)dst_cap;
  } else
  if (flags & PCO_GFX_TASK_HAS_FILE_HEADER) {  // standalone/decompressor.rs:85-137
    const uint32_t magic = (uint32_t)mr.read(32);
    if (!mr.in_bounds()) status = PCO_GFX_INSUFFICIENT_DATA; else if (magic != 0x216f6370u) status = PCO_GFX_CORRUPTION;
    if (!status) {
      const uint32_t sv = (uint32_t)mr.read(8);
      if (sv < 2) mr.bit -= 8;
      else {
        if (sv >= 3) { const uint32_t ub = (uint32_t)mr.read(8); if (ub != 0) { if (dtype_bits(ub) == 0) status = PCO_GFX_CORRUPTION; uniform_type = ub; } }
        if (!status) { const uint32_t power = 1 + (uint32_t)mr.read(kBitsVarintPower); mr.read(power); if (!mr.drain_empty_byte() && mr.in_bounds()) status = PCO_GFX_CORRUPTION; }
      }
      if (!status && !mr.in_bounds()) status = PCO_GFX_INSUFFICIENT_DATA;
      if (!status && sv > 3) status = PCO_GFX_CORRUPTION;
      if (!status) {
        format_major = (uint32_t)mr.read(8);
        if (format_major >= 4) mr.read(8);
        if (format_major > 4) status = PCO_GFX_CORRUPTION; else if (!mr.in_bounds()) status = PCO_GFX_INSUFFICIENT_DATA;
      }
    }
    if (status) { fail(status); return; }
  } else if (src_len == 0) { fail(kStatusRetryLegacy); return; }  // empty stream: zero chunks
  else if ((flags & PCO_GFX_TASK_ONE_CHUNK) && ((flags >> 8) & 0xffu) != 0) format_major = (flags >> 8) & 0xffu;   // the caller read the file's header itself
  if (!wrapped) {   // chunk preamble (standalone/decompressor.rs:190-231)
    const uint32_t tb = (uint32_t)mr.read(8);
    if (!mr.in_bounds()) { fail(PCO_GFX_INSUFFICIENT_DATA); return; }
    if (tb == 0) { fail(kStatusRetryLegacy); return; }  // empty file: the general path reports it
    if ((uniform_type && uniform_type != tb) || tb != dtype) { fail(PCO_GFX_CORRUPTION); return; }
    n = (uint32_t)mr.read(kBitsNEntries) + 1;
    if (!mr.in_bounds()) { fail(PCO_GFX_INSUFFICIENT_DATA); return; }
    if (n > dst_cap) { fail(PCO_GFX_INVALID_ARGUMENT); return; }
  }
  const uint32_t num_kind = dtype_kind(dtype);
  // ChunkMeta (metadata/chunk.rs:127-174), as in decode_kernel.hip::decode_chunk
  const uint32_t mode_kind = (uint32_t)mr.read(kBitsModeVariant);
  L mode_base = 0; uint32_t mode_k = 0;
  if (mode_kind == kIntMult) { if (format_major == 0) { fail(PCO_GFX_CORRUPTION); return; } mode_base = (L)mr.read(LB); }
  else if (mode_kind == kFloatMult) mode_base = (L)mr.read(LB);
  else if (mode_kind == kFloatQuant) mode_k = (uint32_t)mr.read(kBitsQuantK);
  else if (mode_kind == kDict) { fail(mr.in_bounds() ? kStatusRetryLegacy : PCO_GFX_INSUFFICIENT_DATA); return; }   // (the general kernel decodes Dict)
  else if (mode_kind != kClassic) { fail(mr.in_bounds() ? PCO_GFX_CORRUPTION : PCO_GFX_INSUFFICIENT_DATA); return; }
  if (!mr.in_bounds()) { fail(PCO_GFX_INSUFFICIENT_DATA); return; }
  uint32_t dkind = kDeltaNone, dorder = 0, wlog = 0, slog = 0; bool sec_uses_delta = false;
  if (format_major < 3) { dorder = (uint32_t)mr.read(kBitsDeltaOrder); if (dorder) dkind = kDeltaConsecutive; }
  else {
    const uint32_t variant = (uint32_t)mr.read(kBitsDeltaVariant);
    if (variant == 1) { dorder = (uint32_t)mr.read(kBitsDeltaOrder); if (dorder == 0) { fail(PCO_GFX_CORRUPTION); return; } dkind = kDeltaConsecutive; sec_uses_delta = mr.read(1) != 0; }
    else if (variant == 2) {
      wlog = 1 + (uint32_t)mr.read(kBitsLookbackWindowLog); slog = (uint32_t)mr.read(kBitsLookbackStateLog);
      if (wlog > kMaxLookbackWindowLog || slog > wlog) { fail(PCO_GFX_CORRUPTION); return; }
      dkind = kDeltaLookback; sec_uses_delta = mr.read(1) != 0;
    } else if (variant == 3) { fail(mr.in_bounds() ? kStatusRetryLegacy : PCO_GFX_INSUFFICIENT_DATA); return; }   // Conv1: the general kernel
    else if (variant != 0) { fail(PCO_GFX_CORRUPTION); return; }
  }
  if (!mr.in_bounds()) { fail(PCO_GFX_INSUFFICIENT_DATA); return; }
  const uint32_t present[3] = {dkind == kDeltaLookback ? 1u : 0u, 1u, (mode_kind == kIntMult || mode_kind == kFloatMult || mode_kind == kFloatQuant) ? 1u : 0u};
  uint32_t total_tbl = 0; bool too_big = false;
  {
    MetaReader peek = mr;
#pragma unroll
    for (int vi = 0; vi < 3; vi++) {
      VarInfo v{};
      v.present = present[vi]; v.latent_bits = vi == 0 ? 32u : LB;
      if (vi == 1 || (vi == 2 && sec_uses_delta)) { v.delta_kind = dkind; v.delta_order = dorder; v.window_n_log = wlog; v.state_n_log = slog; }
      if (present[vi]) {
        const uint32_t a = (uint32_t)peek.read(kBitsAnsSizeLog), nb = (uint32_t)peek.read(kBitsNBins);
        if (!peek.in_bounds()) { fail(PCO_GFX_INSUFFICIENT_DATA); return; }
        if ((1u << a) < nb || (nb == 1 && a > 0) || a > kMaxAnsBits) { fail(PCO_GFX_CORRUPTION); return; }
        v.ans_size_log = a; v.n_bins = nb;
        v.off_nodes = total_tbl; v.off_lower = 0; v.off_ob = total_tbl + (4u << a);
        total_tbl += (4u << a) + 8u;   // entries + bin 0's offset bits (the other bins' ride in the entries)
        if (nb > kFastMaxBins || a > 12) too_big = true;
        peek.bit += (uint64_t)nb * (a + v.latent_bits + offset_bits_bits(v.latent_bits));
      }
      if (lane == 0) { uint32_t PCO_LDS* p = (uint32_t PCO_LDS*)(vinfo + vi); const uint32_t* qq = (const uint32_t*)&v; for (int w = 0; w < 16; w++) p[w] = qq[w]; }
    }
  }
  wave_sync_lds();
  if (too_big || total_tbl > WalkCfg<4>::kGrpTblBytes) { fail(kStatusRetryLegacy); return; }
  if (total_tbl > kGrpTblBytes) { fail(WalkCfg<KQ>::kRetryStatus); return; }
  if (dkind == kDeltaLookback && (mode_kind != kClassic || sec_uses_delta)) { fail(kStatusRetryLegacy); return; }  // the general path reports it
#pragma unroll
  for (int vi = 0; vi < 3; vi++) {
    if (!present[vi]) continue;
    mr.bit += kBitsAnsSizeLog + kBitsNBins;
    bool ok;
    if constexpr (kInline) ok = vi == 0 ? fast_build_var_impl<uint32_t, KQ>(q, vi, mr, bins_out, status) : fast_build_var_impl<L, KQ>(q, vi, mr, bins_out, status);
    else ok = vi == 0 ? fast_build_var<uint32_t, KQ>(q, vi, mr, bins_out, status) : fast_build_var<L, KQ>(q, vi, mr, bins_out, status);
    if (!ok) { fail(status); return; }
  }
  if (!mr.drain_empty_byte()) { if (mr.in_bounds()) { fail(PCO_GFX_CORRUPTION); return; } }
  if (!mr.in_bounds()) { fail(PCO_GFX_INSUFFICIENT_DATA); return; }
  if (wrapped && meta_p != nullptr) mr = MetaReader{src, src_len, 0};   // the page has a buffer of its own: every bit position from here on is relative to task.src
  if (dkind == kDeltaLookback) {  // metadata/chunk.rs:38-57
    const uint32_t nb0 = uni(vinfo[0].n_bins);
    const uint64_t PCO_GLOBAL* lw = (const uint64_t PCO_GLOBAL*)bins_out;
    uint32_t bad = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    for (uint32_t b = lane; b < nb0; b += 64) { const uint64_t x = __hip_atomic_load(&lw[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (x < 1 || x > (1ull << wlog)) bad = 1; }
    if (uni(wave_or_u32(bad))) { fail(PCO_GFX_CORRUPTION); return; }
  }
  {  // mode validity (data_types/unsigned.rs:65-71, float.rs:377-390)
    bool valid = true;
    if (mode_kind == kIntMult) valid = num_kind != kFloat && mode_base > 0;
    else if (mode_kind == kFloatQuant) { const uint32_t prec = LB == 64 ? 52 : (LB == 32 ? 23 : 10); valid = num_kind == kFloat && mode_k > 0 && mode_k <= prec; }
    else if (mode_kind == kFloatMult) {
      if (num_kind != kFloat) valid = false;
      else if constexpr (sizeof(L) >= 4) { typedef typename FloatOf<L>::F F; const F b = bits_to_float(from_latent_ordered<L>(mode_base, kFloat)); valid = isfinite(b) && b != (F)0; }
      else if constexpr (sizeof(L) == 2) { const uint32_t hb = (uint32_t)from_latent_ordered<L>(mode_base, kFloat); valid = (hb & 0x7c00u) != 0x7c00u && (hb & 0x7fffu) != 0; }
      else valid = false;
    }
    if (!valid) { fail(PCO_GFX_CORRUPTION); return; }
  }
  // page meta (metadata/page.rs:36-57, page_latent_var.rs:28-49)
  uint32_t nlps[3];
#pragma unroll
  for (int vi = 0; vi < 3; vi++) {
    const uint32_t dk = uni(vinfo[vi].delta_kind);
    nlps[vi] = dk == kDeltaConsecutive ? uni(vinfo[vi].delta_order) : (dk == kDeltaLookback ? (1u << uni(vinfo[vi].state_n_log)) : 0u);
  }
  L PCO_GLOBAL* dst = (L PCO_GLOBAL*)task.dst;
#pragma unroll
  for (int vi = 0; vi < 3; vi++) {
    if (!present[vi]) continue;
    const uint32_t lbits = vi == 0 ? 32u : LB;
    const uint32_t dk = uni(vinfo[vi].delta_kind);
    for (uint32_t i = 0; i < nlps[vi]; i++) {
      const L x = (L)mr.read(lbits);
      if (dk == kDeltaConsecutive) { if (lane == 0 && i < 8) plan->moments[vi == 2 ? 1 : 0][i] = (uint64_t)x; if (i < 2) out.moments[vi == 2 ? 1 : 0][i] = (uint64_t)x; }
      else if (vi == 1 && i < n && mr.in_bounds()) { if (lane == 0) dst[i] = from_latent_ordered<L>(x, num_kind); }
    }
    const uint32_t asl = uni(vinfo[vi].ans_size_log);
    for (uint32_t j = 0; j < 4; j++) out.states[vi][j] = (uint32_t)mr.read(asl);
  }
  if (!mr.drain_empty_byte()) { if (mr.in_bounds()) { fail(PCO_GFX_CORRUPTION); return; } }
  if (!mr.in_bounds()) { fail(PCO_GFX_INSUFFICIENT_DATA); return; }
  const uint32_t n_in_body = n > nlps[1] ? n - nlps[1] : 0;
  if (n_in_body > 0) for (int vi = 0; vi < 3; vi++) if (present[vi] && uni(vinfo[vi].n_bins) == 0) { fail(PCO_GFX_CORRUPTION); return; }
  out.mode_kind = mode_kind;
  if (lane == 0) {
    plan->status = PCO_GFX_OK; plan->n = n; plan->mode_kind = mode_kind; plan->mode_k = mode_k; plan->mode_base = (uint64_t)mode_base;
    plan->num_kind = num_kind; plan->dtype = dtype; plan->window_n_log = wlog; plan->state_n_log = slog; plan->consumed = 0; plan->fused = 0; plan->more = 0;
    for (int vi = 0; vi < 3; vi++) {
      plan->present[vi] = present[vi]; plan->n_bins[vi] = vinfo[vi].n_bins; plan->max_ob[vi] = vinfo[vi].max_ob;
      plan->delta_kind[vi] = vinfo[vi].delta_kind; plan->delta_order[vi] = vinfo[vi].delta_order; plan->nlps[vi] = nlps[vi];
    }
  }
  out.n = n; out.bitpos = mr.bit;
}

template <class L, uint32_t KQ>
__device__ __noinline__ void fast_front(const PcoGfxDecodeTask& task, uint32_t q, DecPlan PCO_GLOBAL* plan, uint8_t PCO_GLOBAL* bins_out, FrontOut& out, gcptr_u8 meta_p, uint64_t meta_len) { fast_front_impl<L, KQ, false>(task, q, plan, bins_out, out, meta_p, meta_len); }

// ---------------------------------------------------------------------------------------------------------
// dec_walk_kernel: kWQ chunks per wave, four lanes per chunk (lane 4c+j walks tANS chain j of chunk slot c).
//
// The walk is a serial chain of n/4 steps per chunk and a wave has no other wave to hide behind (the tables fill
// the LDS at one wave per SIMD), so a step costs the entry's LDS latency plus every instruction between its arrival
// and the next ds_read, plus the DS instructions themselves: a lone wave pays 15-30 issue cycles per DS instruction
// that nothing hides (scripts/micro/lds_micro.hip), and the CU's four walker waves share one LDS.  walk_step()
// therefore keeps the dependent stretch to
//   3 x v_and_dpp + v_add3 (row_shr 1,2,3 masked to the quad: bits read by the chains before mine; an entry's low
//   six bits are bits_to_read, so the raw entries are summed and only bits [5:0] of the sum are consumed)
//   -> 2 x v_alignbit (the 64 window bits at the chunk's bit position, from three dwords fetched in the shadow of
//   the previous step) -> v_lshrrev_b64 -> v_bfe (width = entry[4:0]) -> v_lshrrev, v_add_lshl, v_add (next address)
// does everything else (the step's total bit count, the next window fetch, the symbol, the offset-bit count) after
// that ds_read has been issued, and issues three DS instructions per step (entry, ds_read2_b32 + ds_read_b32 for
// the window): the bin's offset-bit count rides in the entry instead of being looked up.  Measured alternatives at
// four waves per CU: a look-ahead window fetch that takes the cut off the dependent stretch needs a byte-granular
// 16-byte read (unaligned ds_read_b128) or four dwords and a three-way cut, and both lose (7.96 / 9.56 vs 6.89 ms);
// a register-resident window advanced by v_cndmask costs 34 VALU per step (11.3 ms).  Symbols leave in groups of 64:
// bytes [16 j + 4 b, +4) of a group hold chain j's symbols of the four steps of the group's block b (one 16-byte store per
// lane and group -- sixteen 4-byte stores per batch backed up the VMEM queue; dec_expand_kernel undoes the layout).
// ---------------------------------------------------------------------------------------------------------
struct WalkRegs {
  uint32_t saddr;                    // LDS byte address of the current state's entry
  uint32_t e;                        // that entry (its load is issued as soon as the address is known)
  uint32_t bitaddr;                  // LDS BIT address (8 x byte address + bit) of the next unread bit
  uint32_t d0, d1, d2;               // the window dwords holding bits [bitaddr, bitaddr + 64)
  uint32_t obsum, symacc;
};

// Advance the bit position by `tot` and fetch the three window dwords at the new position.
__device__ __forceinline__ void walk_window(WalkRegs& r, uint32_t tot) {
  r.bitaddr += tot;
  const uint32_t PCO_LDS* w = (const uint32_t PCO_LDS*)(uintptr_t)((r.bitaddr >> 3) & ~3u);
  r.d0 = w[0]; r.d1 = w[1]; r.d2 = w[2];
}

// sixteen symbol bytes of one lane: one 16-byte store, or -- when a kernel on another XCD reads them while this one runs -- two
// agent-scope 8-byte stores (written through, see decode_trail.hip)
template <bool kAgent>
__device__ __forceinline__ void store_syms(uint8_t PCO_GLOBAL* p, uint32_t __attribute__((ext_vector_type(4))) acc) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#ifdef PCO_TRAIL_PLAINST
  if constexpr (false) {
#else
  if constexpr (kAgent) {
#endif
#ifdef PCO_TRAIL_ST8
    __hip_atomic_store((uint64_t*)p, (uint64_t)acc.x | ((uint64_t)acc.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store((uint64_t*)p + 1, (uint64_t)acc.z | ((uint64_t)acc.w << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    // one 16-byte store with the scope bit the compiler gives an agent-scope atomic store (sc1: written through to where every XCD reads it);
    // there is no 16-byte atomic to ask for, and the reader takes the sixteen bytes as four independent dwords anyway
    __asm__ volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(acc) : "memory");
#endif
  } else *(u32x4 PCO_GLOBAL*)p = acc;
}

struct QuadMasks { uint32_t m1, m2, m3, c63; };   // all-ones where chain j >= 1, 2, 3; 63

template <int K, bool kTail>
__device__ __forceinline__ void walk_step(WalkRegs& r, const QuadMasks& m, uint32_t tbl_addr, bool chain_on) {
  uint32_t e = r.e;
  if (kTail) e = chain_on ? e : 0u;   // an exhausted chain reads no bits and keeps its state
  // ---- critical path ----
  const uint32_t p = ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)e, 0x111, 0xf, 0xf, true) & m.m1) +   // row_shr:1, chains 1..3
                     ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)e, 0x112, 0xf, 0xf, true) & m.m2) +   // row_shr:2, chains 2..3
                     ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)e, 0x113, 0xf, 0xf, true) & m.m3);    // row_shr:3, chain 3
  const uint32_t x0 = __builtin_amdgcn_alignbit(r.d1, r.d0, r.bitaddr), x1 = __builtin_amdgcn_alignbit(r.d2, r.d1, r.bitaddr);   // 64 bits from the position
  const uint64_t xs = (((uint64_t)x1 << 32) | x0) >> (p & 63u);
  const uint32_t v = __builtin_amdgcn_ubfe((uint32_t)xs, 0u, e);                                 // width = e[4:0]
  const uint32_t next = tbl_addr + (((e >> 21) + v) << 2);
  if (kTail) r.saddr = chain_on ? next : r.saddr; else r.saddr = next;
  r.e = *(const uint32_t PCO_LDS*)(uintptr_t)r.saddr;
  __builtin_amdgcn_sched_barrier(0);
  // ---- in the shadow of that load ----
  const uint32_t t = p + e;                                                                       // bits [5:0]: through my chain
  walk_window(r, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0xFF, 0xf, 0xf, true) & m.c63);  // quad_perm [3,3,3,3]: the step's total
  if (K == 0) r.symacc = __builtin_amdgcn_ubfe(e, 13u, 8u); else r.symacc |= __builtin_amdgcn_ubfe(e, 13u, 8u) << (8 * K);
  r.obsum += __builtin_amdgcn_ubfe(e, 6u, 7u);   // (an exhausted chain's e is 0)
  __builtin_amdgcn_sched_barrier(0);   // keep the shadow work ahead of the next step's wait for the entry
}

#ifdef PCO_WALK_TIMING
__device__ unsigned long long g_walk_timing[8];
#define WT_NOW() __builtin_readcyclecounter()
#endif
// accept_status: 0 = the first stage (every task), else only the tasks an earlier stage left with that status.
// kTrail (kWQ == 8, first stage only): the chunks the trailing expanders can take (no lookback, at most 64 bins per variable, delta
// orders up to 2 on the primary variable only) are marked DecPlan::fused and their progress is published batch by batch (decode_trail.hip);
// everything the expanders read -- symbols, section starts, progress -- leaves through agent-scope stores, the plans and bins through
// one agent-scope release after the table build.
// Does any of the (up to eight) chunks of walker block `wb` look like one the trailing expanders take?  A glance at the chunk preamble
// (standalone/decompressor.rs:190-231: type byte, 24-bit count, then ChunkMeta's 4 mode bits and 4 delta bits): a plain one-chunk task in
// classic mode without lookback / Conv1.  The two first-stage walkers split the blocks by this -- the one that publishes its progress pays
// ~0.7 us a round for it (agent-scope stores, a wait), which a call of float-mult or lookback chunks should not.  The precise test (bins
// that fit a wave's registers, delta order <= 2) is the publishing walker's, after it has parsed the metadata.
// Returns bit 0: a one-variable candidate (classic mode), bit 1: a two-variable candidate (int-mult / float-mult / float-quant: the blocks
// dec_trail_kernel<L, true> follows).
__device__ __forceinline__ uint32_t block_trail_kinds(const PcoGfxDecodeTask* tasks, const uint32_t* task_ids, uint32_t n_ids, uint32_t wb, const MetaRef* metas) {
  const uint32_t lane = lane_id(), bi = wb * 8 + (lane & 7u);
  bool one = false, two = false;
  if (lane < 8 && bi < n_ids) {
    const uint32_t ti = task_ids ? task_ids[bi] : bi;
    const PcoGfxDecodeTask t = tasks[ti];
    const bool wrapped = (t.flags & PCO_GFX_TASK_WRAPPED_PAGE) != 0;
    const uint32_t fmt = wrapped ? (t.flags >> 8) & 0xffu : ((t.flags & PCO_GFX_TASK_ONE_CHUNK) && ((t.flags >> 8) & 0xffu) ? (t.flags >> 8) & 0xffu : 4u);
    // where the ChunkMeta starts: behind the 4-byte chunk preamble of a standalone chunk, at the first byte of a wrapped page's metadata
    gcptr_u8 mp = (gcptr_u8)t.src + 4; uint64_t mlen = t.src_len >= 4 ? t.src_len - 4 : 0;
    if (wrapped) { if (metas != nullptr && metas[ti].p != nullptr) { mp = (gcptr_u8)metas[ti].p; mlen = metas[ti].len; } else { mp = (gcptr_u8)t.src; mlen = t.src_len; } }
    if ((t.flags & (PCO_GFX_TASK_META_ONLY | PCO_GFX_TASK_HAS_FILE_HEADER)) == 0 && mlen >= 20 && fmt >= 3 && fmt <= 4) {
      const uint64_t w = load_u64_le(mp);   // mode (4 bits) | [base (the type's bits) or k (8)] | delta variant (4) | [order (3) | secondary uses delta (1)] | ans_size_log (4) | n_bins (15)
      const uint32_t mode = (uint32_t)w & 15u;
      if (mode == kClassic) {
        const uint32_t dv = (uint32_t)(w >> 4) & 15u;
        const uint32_t n_bins = (uint32_t)(w >> (dv == kDeltaConsecutive ? 16 : 12)) & 0x7fffu;
        one = (dv == kDeltaNone || dv == kDeltaConsecutive) && n_bins > 1 && n_bins <= kTrailMaxBins;   // (one bin: nothing to walk, nothing to hide the expansion under)
      } else if (mode == kIntMult || mode == kFloatMult || mode == kFloatQuant) {
        const uint32_t at = 4u + (mode == kFloatQuant ? kBitsQuantK : (uint32_t)dtype_bits(t.dtype));   // where the delta variant starts, in bits from byte 4
        const uint64_t w2 = load_u64_le(mp + (at >> 3));
        const uint32_t dv = (uint32_t)(w2 >> (at & 7u)) & 15u;
        two = dv == kDeltaNone || dv == kDeltaConsecutive;
      }
    }
  }
  return (uni((uint32_t)__any(one)) ? 1u : 0u) | (uni((uint32_t)__any(two)) ? 2u : 0u);
}
__device__ __forceinline__ bool block_has_trail_candidate(const PcoGfxDecodeTask* tasks, const uint32_t* task_ids, uint32_t n_ids, uint32_t wb, const MetaRef* metas) {
  return block_trail_kinds(tasks, task_ids, n_ids, wb, metas) != 0;
}

#ifdef PCO_TRAIL_TIMING   // (measurement builds: when every walker block and every expander block started and ended, on the device-wide 100 MHz clock)
constexpr uint32_t kTrailStampBlocks = 4096;
__device__ unsigned long long g_trail_stamps[4][kTrailStampBlocks];   // walker start, walker end, expanders start, expanders end
#define PCO_TRAIL_STAMP(which, idx) do { if (lane_id() == 0 && (idx) < kTrailStampBlocks) g_trail_stamps[which][idx] = wall_clock64(); } while (0)
#else
#define PCO_TRAIL_STAMP(which, idx) do { } while (0)
#endif
template <class L, uint32_t kWQ, bool kTrail>
__device__ __forceinline__ void dec_walk_body(const PcoGfxDecodeTask* tasks, const uint32_t* task_ids, uint32_t n_ids, DecPlan* plans,
                                              uint8_t* bins_area, uint8_t* sym_area, uint64_t sym_stride, uint64_t* offpos_area, uint64_t offpos_stride,
                                              uint32_t accept_status, PcoGfxTaskResult* results, uint32_t* progress, const MetaRef* metas) {
  static_assert(!kTrail || kWQ == 8, "the trailing expanders follow the eight-chunk walker");
  const uint32_t wb = walk_block_id();   // (the wave's "block": blockIdx.x in the one-wave kernels)
  if ((uint64_t)wb * kWQ >= n_ids) return;   // (the spare waves of the last four-wave workgroup)
  constexpr uint32_t kGrpBytes = WalkCfg<kWQ>::kGrpBytes;
#ifndef PCO_TRAIL_NOPRIO
  if constexpr (kTrail) __builtin_amdgcn_s_setprio(3);   // the walker's chain sets the duration of the decode: it goes first whenever it can issue
#endif
  const uint32_t lane = lane_id();
  const uint32_t slot = lane >> 2, j = lane & 3;
  if constexpr (kTrail) {   // the blocks without a candidate belong to the ordinary walker (launched beside this one); their expanders are told at once
    if (!block_has_trail_candidate(tasks, task_ids, n_ids, wb, metas)) {
      if (lane < 8) __hip_atomic_store(progress + (uint64_t)wb * kTrailProgressStride + lane, kTrailDead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
  } else if constexpr (kWQ == 8) {
    if (progress != nullptr && accept_status == 0 && block_has_trail_candidate(tasks, task_ids, n_ids, wb, metas)) return;   // (progress != nullptr: the publishing walker runs too and takes this block)
  }
  if constexpr (kTrail) PCO_TRAIL_STAMP(0, wb);
  // ---- phase 0: metadata + tables, one task at a time with the whole wave; slot q belongs to lanes 4q..4q+3 ----
  uint32_t my_ti = 0xffffffffu, my_active = 0, my_front_ok = 0, my_n = 0, my_flags = 0, my_mode = kClassic;
  uint32_t st0 = 0, st1 = 0, st2 = 0;   // this lane's chain state per variable, as an entry address
  uint64_t my_bitpos = 0, my_len = 0;
  uint64_t my_mom[2][2] = {{0, 0}, {0, 0}};
  gcptr_u8 my_src = nullptr;
  for (uint32_t q = 0; q < kWQ; q++) {
    const uint32_t bi = wb * kWQ + q;
    if (bi >= n_ids) break;
    const uint32_t ti = task_ids ? task_ids[bi] : bi;
    if (accept_status != 0 && uni(((const DecPlan PCO_GLOBAL*)plans + ti)->status) != accept_status) continue;   // an earlier stage dealt with this task
    const PcoGfxDecodeTask task = tasks[ti];
    FrontOut fo;
    gcptr_u8 meta_p = metas != nullptr ? (gcptr_u8)metas[ti].p : (gcptr_u8) nullptr;
    const uint64_t meta_len = metas != nullptr ? uni((uint64_t)metas[ti].len) : 0;
    if constexpr (kTrail) fast_front_impl<L, kWQ, true>(task, q, (DecPlan PCO_GLOBAL*)plans + ti, (uint8_t PCO_GLOBAL*)bins_area + (uint64_t)ti * kBinsAreaPerTask, fo, meta_p, meta_len);
    else fast_front<L, kWQ>(task, q, (DecPlan PCO_GLOBAL*)plans + ti, (uint8_t PCO_GLOBAL*)bins_area + (uint64_t)ti * kBinsAreaPerTask, fo, meta_p, meta_len);
    if (slot == q) {
      my_ti = ti; my_active = fo.status == PCO_GFX_OK ? 1u : 0u; my_front_ok = my_active; my_n = fo.n; my_bitpos = fo.bitpos; my_mode = fo.mode_kind;
      my_len = task.src_len; my_flags = task.flags; my_src = (gcptr_u8)task.src;
      my_mom[0][0] = fo.moments[0][0]; my_mom[0][1] = fo.moments[0][1]; my_mom[1][0] = fo.moments[1][0]; my_mom[1][1] = fo.moments[1][1];
      st0 = j == 0 ? fo.states[0][0] : (j == 1 ? fo.states[0][1] : (j == 2 ? fo.states[0][2] : fo.states[0][3]));
      st1 = j == 0 ? fo.states[1][0] : (j == 1 ? fo.states[1][1] : (j == 2 ? fo.states[1][2] : fo.states[1][3]));
      st2 = j == 0 ? fo.states[2][0] : (j == 1 ? fo.states[2][1] : (j == 2 ? fo.states[2][2] : fo.states[2][3]));
    }
    wave_sync_lds();
  }
  const uint32_t slice = (slot < kWQ ? slot : 0u) * kGrpBytes;   // LDS byte offset of this chunk's slice
  uint8_t PCO_LDS* gbase = walk_lds<kWQ>() + slice;
  const uint32_t lds0 = (uint32_t)(uintptr_t)walk_lds<kWQ>();
  uint64_t PCO_LDS* win = (uint64_t PCO_LDS*)(gbase + kGrpWinOff);
  VarInfo PCO_LDS* vinfo = (VarInfo PCO_LDS*)(gbase + kGrpVarOff);
  // ---- phase 1: rounds; in each round every active chunk handles its next (batch, variable) item ----
  uint32_t n_rem = my_n, batch = 0, cur_v = 0, status = PCO_GFX_OK, present_mask = 0, nlps1 = 0;
  if (my_active) {
    present_mask = (vinfo[0].present ? 1u : 0u) | 2u | (vinfo[2].present ? 4u : 0u);
    nlps1 = vinfo[1].delta_kind == kDeltaConsecutive ? vinfo[1].delta_order : (vinfo[1].delta_kind == kDeltaLookback ? (1u << vinfo[1].state_n_log) : 0u);
    cur_v = (present_mask & 1u) ? 0u : 1u;
    if (n_rem == 0) my_active = 0;
    // states -> entry addresses
    st0 = lds0 + slice + kGrpTblOff + vinfo[0].off_nodes + 4u * st0;
    st1 = lds0 + slice + kGrpTblOff + vinfo[1].off_nodes + 4u * st1;
    st2 = lds0 + slice + kGrpTblOff + vinfo[2].off_nodes + 4u * st2;
  }
  bool my_fused = false;
  uint32_t my_fused_kind = 0;   // DecPlan::fused: 1 = one latent variable, 2 = two (dec_trail_kernel<L, true>)
  uint32_t PCO_GLOBAL* my_progress = nullptr;
  if constexpr (kTrail) {
    // which of the wave's chunks the trailing expanders take: no lookback variable, every variable's bins in one wave's registers (<= 64),
    // offsets of up to 16 bits (a lane's four fields in one 64-bit window), delta orders up to 2 on the primary variable (moments in
    // registers) and none on the secondary.  One variable (classic mode): 2..64 bins -- with one bin there is nothing to walk and nothing to
    // hide the expansion under.  Two (int-mult, float-mult, float-quant): the expanders of the second kind.  The others go to dec_expand_kernel.
    if (my_active && slot < kWQ) {
      const bool prim_ok = vinfo[0].present == 0 && vinfo[1].n_bins >= 1 && vinfo[1].n_bins <= kTrailMaxBins && vinfo[1].max_ob <= 16 &&
                           (vinfo[1].delta_kind == kDeltaConsecutive ? vinfo[1].delta_order <= 2 : vinfo[1].delta_kind == kDeltaNone);
      if (vinfo[2].present == 0) { if (prim_ok && my_mode == kClassic && vinfo[1].n_bins > 1) my_fused_kind = 1; }
      else if (prim_ok && (my_mode == kIntMult || my_mode == kFloatMult || my_mode == kFloatQuant) && vinfo[2].n_bins >= 1 && vinfo[2].n_bins <= kTrailMaxBins &&
               vinfo[2].max_ob <= 16 && vinfo[2].delta_kind == kDeltaNone && (vinfo[1].n_bins > 1 || vinfo[2].n_bins > 1)) my_fused_kind = 2;
      my_fused = my_fused_kind != 0;
    }
    if (slot < kWQ) my_progress = (uint32_t PCO_GLOBAL*)progress + (uint64_t)wb * kTrailProgressStride + slot;
    if (j == 0 && my_fused) ((DecPlan PCO_GLOBAL*)plans + my_ti)->fused = my_fused_kind;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // plans and bins (plain stores of the table build) are visible to every XCD from here on
    if (j == 0 && slot < kWQ) __hip_atomic_store((uint32_t*)my_progress, my_fused ? 1u : kTrailDead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }

  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  QuadMasks qm = {j >= 1 ? ~0u : 0u, j >= 2 ? ~0u : 0u, j >= 3 ? ~0u : 0u, 63u};
  asm volatile("" : "+v"(qm.m1), "+v"(qm.m2), "+v"(qm.m3), "+v"(qm.c63));   // opaque, so that they stay VGPR operands of v_and_b32_dpp
  // kTrail: the agent-scope stores (written through: their acknowledgement takes microseconds) of a round are issued at the NEXT round's
  // staging point and have that whole round to complete -- issued where the symbols are produced, the last of them would be waited for
  // together with the next round's staging loads, on the chain (10.3 ms per launch instead of 6.5).
  u32x4 d_acc[4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
  uint32_t d_ngrp = 0, issued_batches = 0; uint8_t PCO_GLOBAL* d_sym_out = nullptr;
  uint64_t d_off_val = 0; uint64_t* d_off_ptr = nullptr;
  auto flush_deferred = [&]() {
#pragma unroll
    for (int g = 0; g < 4; g++) if ((uint32_t)g < d_ngrp) store_syms<true>(d_sym_out + 64 * g, d_acc[g]);
    d_ngrp = 0;
    if (d_off_ptr) __hip_atomic_store(d_off_ptr, d_off_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    d_off_ptr = nullptr;
  };
#ifdef PCO_WALK_TIMING
  unsigned long long wt_stage = 0, wt_walk = 0, wt_tail = 0, wt_rounds = 0, wt_t0 = WT_NOW(), wt_start = wt_t0, wt_s1 = 0, wt_s2 = 0, wt_s3 = 0, wt_t1 = wt_t0;
#endif
#ifndef PCO_WALK_NO_TOUCH
  uint64_t touch_prev = my_bitpos >> 3; uint32_t touch_r0 = 0, touch_r1 = 0;
#endif
  while (__any(my_active != 0)) {
    uint32_t cnt = 0, nb = 0, asl = 0, off_ob = 0, off_nodes = 0;
    bool walk = false;
    if (my_active) {
      const VarInfo PCO_LDS* vi = vinfo + cur_v;
      nb = vi->n_bins; asl = vi->ans_size_log; off_ob = vi->off_ob; off_nodes = vi->off_nodes;
      const uint32_t batch_n = n_rem < kBatchN ? n_rem : kBatchN;
      if (cur_v == 0) { const uint32_t lim = n_rem > nlps1 ? n_rem - nlps1 : 0; cnt = lim < batch_n ? lim : batch_n; }
      else {
        const uint32_t nl = cur_v == 1 ? nlps1 : (vi->delta_kind == kDeltaConsecutive ? vi->delta_order : (vi->delta_kind == kDeltaLookback ? (1u << vi->state_n_log) : 0u));
        const uint32_t rem = n_rem > nl ? n_rem - nl : 0; cnt = rem < kBatchN ? rem : kBatchN;
      }
      walk = cnt > 0 && nb > 1;
    }
    const uint64_t q0 = my_bitpos >> 6;   // first qword of the window, in qwords from src
#ifdef PCO_WALK_TIMING
    { const unsigned long long t = WT_NOW(); wt_s1 += t - wt_t0; wt_t1 = t; }
#endif
    WalkRegs r;
    r.saddr = cur_v == 0 ? st0 : (cur_v == 1 ? st1 : st2);
    const uint32_t win_addr = lds0 + slice + kGrpWinOff, rel0 = (uint32_t)(my_bitpos & 63);
    r.bitaddr = 8u * win_addr + rel0; r.obsum = 0; r.symacc = 0; r.e = 0; r.d0 = r.d1 = r.d2 = 0;
    if (walk) {  // stage the batch's ANS section: the chunk's 4 lanes copy qword pairs, all loads in flight at once
      const uint32_t nq = ((((rel0 + cnt * asl + 63u) >> 6) + 3u) + 1u) & ~1u;   // even, <= 54 of the window's 56 qwords
      uint64_t lo[7], hi[7];
      if ((q0 + nq) * 8 <= my_len + 16) {   // the whole section is inside the buffer
#pragma unroll
        for (int k = 0; k < 7; k++) {
          const uint32_t qi = 2 * j + 8 * k, qc = qi < nq ? qi : nq - 2;   // no branch: the loads issue back to back
          lo[k] = load_u64_le(my_src + (q0 + qc) * 8); hi[k] = load_u64_le(my_src + (q0 + qc) * 8 + 8);
        }
      } else {
#pragma unroll
        for (int k = 0; k < 7; k++) {
          const uint32_t qi = 2 * j + 8 * k;
          lo[k] = 0; hi[k] = 0;
          if (qi < nq) { lo[k] = load_u64_le_safe(my_src, (q0 + qi) * 8, my_len + 16); hi[k] = load_u64_le_safe(my_src, (q0 + qi + 1) * 8, my_len + 16); }
        }
      }
#ifdef PCO_WALK_TIMING
      { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); const unsigned long long t = WT_NOW(); wt_s2 += t - wt_t1; wt_t1 = t; }
#endif
#pragma unroll
      for (int k = 0; k < 7; k++) { const uint32_t qi = 2 * j + 8 * k; if (qi < nq) { win[qi] = lo[k]; win[qi + 1] = hi[k]; } }
    }
    wave_sync_lds();
    if constexpr (kTrail) {
      // everything issued before this round's staging loads is acknowledged (the wave has just waited for those loads): the stores of the
      // round before last, issued at the last staging point, are out -- `issued_batches` batches of my chunk can be published; then the
      // last round's stores go out
#ifndef PCO_TRAIL_NOWAIT
      __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      if (my_fused && j == 0 && status == PCO_GFX_OK) __hip_atomic_store((uint32_t*)my_progress, 1u + issued_batches, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if constexpr (kTrailDefer) { flush_deferred(); issued_batches = batch; }
      else issued_batches = batch;   // (stores issued where they are produced: everything through the last round is out)
    }
#ifdef PCO_WALK_TIMING
    { const unsigned long long t = WT_NOW(); wt_s3 += t - wt_t1; }
#endif
#ifndef PCO_WALK_NO_TOUCH
    // Warm the L2 with the lines this chunk's NEXT round will stage (its position is only known after this walk; the staging
    // loads otherwise wait ~1600 cycles on HBM every round): predicted start = this start + the previous stride, and the
    // chunk's four lanes touch eight 128-byte lines around it.  The loaded values get their (dummy) use one round later,
    // right here, where this round's staging loads have already been waited for -- VMEM returns in order, so that use
    // never waits.
    ((uint32_t PCO_LDS*)(walk_lds<kWQ>() + WalkCfg<kWQ>::kWalkTmpOff))[lane] = touch_r0 ^ touch_r1;   // (the table-build scratch is idle during the walk)
    if (walk) {
      const uint64_t cur = q0 * 8, pred = cur + (cur - touch_prev);
      touch_prev = cur;
      const uint64_t line0 = (pred & ~(uint64_t)127) - (pred >= 128 ? 128u : 0u);
      const uint64_t a0 = line0 + 128u * j, a1 = a0 + 512u;
      touch_r0 = load_u32_le(my_src + (a0 + 4 <= my_len ? a0 : 0));
      touch_r1 = load_u32_le(my_src + (a1 + 4 <= my_len ? a1 : 0));
    }
#endif
#ifdef PCO_WALK_TIMING
    { const unsigned long long t = WT_NOW(); wt_stage += t - wt_t0; wt_t0 = t; }
#endif
    const uint32_t obs_addr = lds0 + slice + kGrpTblOff + off_ob, tbl_addr = lds0 + slice + kGrpTblOff + off_nodes;
    uint8_t PCO_GLOBAL* sym_out = (uint8_t PCO_GLOBAL*)sym_area + ((uint64_t)(my_ti == 0xffffffffu ? 0u : my_ti) * 3 + cur_v) * sym_stride + (uint64_t)batch * kBatchN + 16 * j;
    if (walk) {
      r.e = *(const uint32_t PCO_LDS*)(uintptr_t)r.saddr;
      walk_window(r, 0u);
      constexpr bool defer = kTrail && kTrailDefer;
      if constexpr (defer) d_sym_out = sym_out;
      if (__all(!walk || cnt == kBatchN)) {   // (lanes outside `walk` are masked off here anyway)
#pragma unroll
        for (uint32_t grp = 0; grp < 4; grp++) {   // (unrolled: the trailing form keeps the four groups in registers until the next round)
          u32x4 acc;
#pragma unroll
          for (int b = 0; b < 4; b++) {
            walk_step<0, false>(r, qm, tbl_addr, true);
            walk_step<1, false>(r, qm, tbl_addr, true);
            walk_step<2, false>(r, qm, tbl_addr, true);
            walk_step<3, false>(r, qm, tbl_addr, true);
            acc[b] = r.symacc;
          }
          if constexpr (defer) d_acc[grp] = acc; else store_syms<kTrail>(sym_out + 64 * grp, acc);
        }
        if constexpr (defer) d_ngrp = 4;
      } else {
        const uint32_t steps = (cnt + 3) >> 2;
        if constexpr (defer) d_ngrp = (steps + 15) >> 4;
#pragma unroll
        for (uint32_t grp = 0; grp < 4; grp++) {
          if (grp * 16 >= steps) break;
          u32x4 acc = {0u, 0u, 0u, 0u};
#pragma unroll
          for (int b = 0; b < 4; b++) {
            const uint32_t g = grp * 16 + b * 4;
            if (g < steps) {
              r.symacc = 0;
              if (g + 0 < steps) walk_step<0, true>(r, qm, tbl_addr, 4 * (g + 0) + j < cnt);
              if (g + 1 < steps) walk_step<1, true>(r, qm, tbl_addr, 4 * (g + 1) + j < cnt);
              if (g + 2 < steps) walk_step<2, true>(r, qm, tbl_addr, 4 * (g + 2) + j < cnt);
              if (g + 3 < steps) walk_step<3, true>(r, qm, tbl_addr, 4 * (g + 3) + j < cnt);
              acc[b] = r.symacc;
            }
          }
          if constexpr (defer) d_acc[grp] = acc; else store_syms<kTrail>(sym_out + 64 * grp, acc);
        }
      }
      if (cur_v == 0) st0 = r.saddr; else if (cur_v == 1) st1 = r.saddr; else st2 = r.saddr;
    }
#ifdef PCO_WALK_TIMING
    { const unsigned long long t = WT_NOW(); wt_walk += t - wt_t0; wt_t0 = t; }
#endif
    if (my_active) {
      // the four chains' offset-bit sums -> the chunk total (quad butterfly)
      uint32_t obq = r.obsum;
      obq += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)obq, 0xB1, 0xf, 0xf, false);
      obq += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)obq, 0x4E, 0xf, 0xf, false);
      const uint64_t ans_end = walk ? (q0 << 6) + (r.bitaddr - 8u * win_addr) : my_bitpos;
      uint64_t ob_total = 0;
      if (cnt > 0) ob_total = walk ? (uint64_t)obq : (nb == 1 ? (uint64_t)cnt * *(const uint8_t PCO_LDS*)(uintptr_t)obs_addr : 0ull);
      if (cnt > 0 && j == 0) {
        uint64_t* op = offpos_area + ((uint64_t)my_ti * 3 + cur_v) * offpos_stride + batch;
        if constexpr (kTrail && kTrailDefer) { d_off_ptr = op; d_off_val = ans_end; }
        else if constexpr (kTrail) __hip_atomic_store(op, ans_end, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *op = ans_end;
      }
      my_bitpos = ans_end + ob_total;
      if (my_bitpos > my_len * 8) { status = PCO_GFX_INSUFFICIENT_DATA; my_active = 0; }
      uint32_t nv = cur_v + 1;
      while (nv < 3 && !((present_mask >> nv) & 1u)) nv++;
      if (nv >= 3) {
        const uint32_t batch_n = n_rem < kBatchN ? n_rem : kBatchN;
        n_rem -= batch_n; batch++;
        nv = (present_mask & 1u) ? 0u : 1u;
        if (n_rem == 0) my_active = 0;
      }
      cur_v = nv;
    }
    wave_sync_lds();
#ifdef PCO_WALK_TIMING
    { const unsigned long long t = WT_NOW(); wt_tail += t - wt_t0; wt_t0 = t; wt_rounds++; }
#endif
  }
#ifdef PCO_WALK_TIMING
  if (wb == 0 && lane == 0 && wt_rounds > 0) { g_walk_timing[0] = wt_stage; g_walk_timing[1] = wt_walk; g_walk_timing[2] = wt_tail; g_walk_timing[3] = wt_rounds; g_walk_timing[4] = wt_start; g_walk_timing[5] = WT_NOW(); g_walk_timing[6] = wt_s1; g_walk_timing[7] = wt_s2 | (wt_s3 << 32); }
#endif
  if constexpr (kTrail) {
    flush_deferred();
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (my_fused && j == 0) __hip_atomic_store((uint32_t*)my_progress, status == PCO_GFX_OK ? 1u + batch : kTrailDead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    PCO_TRAIL_STAMP(1, wb);
  }
  // ---- page end (page_decompressor.rs:184-188) and stream end ----
  if (my_ti != 0xffffffffu && j == 0 && slot < kWQ) {
    DecPlan PCO_GLOBAL* plan = (DecPlan PCO_GLOBAL*)plans + my_ti;
    if (my_front_ok) {
      uint64_t bit = my_bitpos;
      if (status == PCO_GFX_OK) {
        const uint32_t sh = (uint32_t)(bit & 7);
        if (sh) { const uint64_t byte = bit >> 3; const uint32_t b = byte < my_len ? my_src[byte] : 0u; if ((b >> sh) != 0) status = PCO_GFX_CORRUPTION; bit += 8 - sh; }
        if (bit > my_len * 8) status = PCO_GFX_INSUFFICIENT_DATA;
      }
      if ((my_flags & PCO_GFX_TASK_WRAPPED_PAGE) && status != PCO_GFX_OK) status = kStatusRetryLegacy;   // (the single-kernel decoder reports how far the page got: see fast_front_impl)
      if (status == PCO_GFX_OK) {
        uint64_t byte = bit >> 3;
        if (my_flags & PCO_GFX_TASK_WRAPPED_PAGE) { }   // a page ends where its last batch ends (+ padding); what follows is the caller's (PageDecompressor::into_src)
        else if (my_flags & PCO_GFX_TASK_ONE_CHUNK) {   // this chunk only: say whether another follows (the caller comes back for it)
          if (byte < my_len && my_src[byte] != 0) plan->more = 1u;
          else if (byte < my_len) { byte += 1; plan->more = 2u; }     // the terminator (aux bit 1: it was there and is consumed)
          else if (my_flags & PCO_GFX_TASK_HAS_FILE_HEADER) status = PCO_GFX_INSUFFICIENT_DATA;
        } else if (my_flags & PCO_GFX_TASK_HAS_FILE_HEADER) {
          if (byte >= my_len) status = PCO_GFX_INSUFFICIENT_DATA;
          else if (my_src[byte] != 0) status = kStatusRetryLegacy;  // another chunk follows: general path
          else byte += 1;
        } else if (byte < my_len) status = kStatusRetryLegacy;
        plan->consumed = byte;
      }
      plan->status = status;
      if constexpr (kTrail) {
        // a chunk the trailing expanders took is finished when they are (the host joins the two streams): its result is written here (a
        // stream with another chunk behind this one goes to the single-kernel decoder whole, which then reports it)
        if (my_fused && status != kStatusRetryLegacy) {
          PcoGfxTaskResult r; r.n_out = status == PCO_GFX_OK ? my_n : 0; r.consumed = plan->consumed; r.status = status; r.aux = plan->more;
          results[my_ti] = r;
        }
      }
    }
  }
}

// split != nullptr (first stage only): the publishing walker runs beside this one and takes the blocks with a candidate for the trailing expanders
// (four walker waves per workgroup like the publishing walker below: one per SIMD by construction)
template <class L, uint32_t kWQ>
__global__ __launch_bounds__(256) void dec_walk_kernel(const PcoGfxDecodeTask* tasks, const uint32_t* task_ids, uint32_t n_ids, DecPlan* plans,
                                                      uint8_t* bins_area, uint8_t* sym_area, uint64_t sym_stride, uint64_t* offpos_area, uint64_t offpos_stride,
                                                      uint32_t accept_status, PcoGfxTaskResult* results, uint32_t* split, const MetaRef* metas) {
  dec_walk_body<L, kWQ, false>(tasks, task_ids, n_ids, plans, bins_area, sym_area, sym_stride, offpos_area, offpos_stride, accept_status, results, split, metas);
}
// The walker that publishes its progress to the trailing expanders: at most 128 VGPRs (four waves per SIMD's worth -- what the ordinary
// walker takes), so that two expander waves of up to 192 fit beside it on every SIMD.  Tighter caps do not pay: at 80 or 96 the per-round
// code (the staging loads' fourteen 64-bit pieces) spills, and a scratch reload on the chain costs more than anything the cap buys
// (10.8 ms per launch against 7.2).
#define PCO_TRAIL_WALK_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))
template <class L>
__global__ __launch_bounds__(256) PCO_TRAIL_WALK_ATTR void dec_walk_trail_kernel(const PcoGfxDecodeTask* tasks, const uint32_t* task_ids, uint32_t n_ids, DecPlan* plans,
                                                      uint8_t* bins_area, uint8_t* sym_area, uint64_t sym_stride, uint64_t* offpos_area, uint64_t offpos_stride,
                                                      PcoGfxTaskResult* results, uint32_t* progress, const MetaRef* metas) {
  dec_walk_body<L, 8, true>(tasks, task_ids, n_ids, plans, bins_area, sym_area, sym_stride, offpos_area, offpos_stride, 0u, results, progress, metas);
}

// A full batch of 64-bit numbers, four per lane in element order, stored as two instructions that each cover one contiguous KB (lane l:
// bytes [16 l, 16 l + 16) of the KB).  Written where they sit -- 32 bytes per lane, two 16-byte stores at a 32-byte lane stride -- every
// store instruction touches half of each line it reaches: twice the write requests per byte, 3.4 TB/s of output where contiguous
// instructions reach 6-7 (scripts/micro/bw_test.hip), and the walker's own loads queue behind them.  The 16-byte units 2 m (numbers 4 m,
// 4 m + 1) and 2 m + 1 (4 m + 2, 4 m + 3) sit in lane m; v_permlane32_swap puts the units of lanes 0-31 into one register set (even units in
// lanes 0-31, odd ones in lanes 32-63) and those of lanes 32-63 into the other, and one crossbar read per dword brings unit l to lane l.
__device__ __forceinline__ void store_u64_batch(unsigned long long PCO_GLOBAL* batch_base, const unsigned long long (&x)[4]) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const uint32_t lane = lane_id();
  const uint32_t from = 4u * (((lane & 1u) << 5) + (lane >> 1));   // (byte address of the source lane for the crossbar)
  const uint32_t a[4] = {(uint32_t)x[0], (uint32_t)(x[0] >> 32), (uint32_t)x[1], (uint32_t)(x[1] >> 32)};
  const uint32_t b[4] = {(uint32_t)x[2], (uint32_t)(x[2] >> 32), (uint32_t)x[3], (uint32_t)(x[3] >> 32)};
  u32x4 s1, s2;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const auto r = __builtin_amdgcn_permlane32_swap(a[j], b[j], false, false);   // r[0] = {a lanes 0-31, b lanes 0-31}, r[1] = {a lanes 32-63, b lanes 32-63}
    s1[j] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)from, (int)r[0]); s2[j] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)from, (int)r[1]);
  }
  u32x4 PCO_GLOBAL* o = (u32x4 PCO_GLOBAL*)batch_base + lane;
  // (non-temporal: nobody on this device reads the numbers back, and kept out of the L2 they leave the walker's prefetched lines alone --
  //  11.1 -> 10.4 ms per 8192 chunks)
#ifdef PCO_DEC_PLAINSTORE
  o[0] = s1; o[64] = s2;
#else
  __builtin_nontemporal_store(s1, o); __builtin_nontemporal_store(s2, o + 64);
#endif
}

// ---------------------------------------------------------------------------------------------------------
// dec_expand_kernel: one wave per chunk, everything after the tANS walk
// ---------------------------------------------------------------------------------------------------------
constexpr uint32_t kExpWaves = 4;                          // waves per chunk: batch b is unpacked by wave b % 4
constexpr uint32_t kExpLowOff = 0;                         // u64[3][256] lowers
constexpr uint32_t kExpObOff = 3 * 2048;                   // u8[3][256]
constexpr uint32_t kExpMomOff = kExpObOff + 768;           // u64[2][8] delta moments, carried from batch to batch
constexpr uint32_t kExpTurnOff = kExpMomOff + 128;         // u32 turn (next batch allowed into the ordered section), u32 lookback-oob flag
constexpr uint32_t kExpWaveOff = kExpTurnOff + 16;         // per wave: dlat u32[256] | scratch u64[256] | parent u32[256]
constexpr uint32_t kExpWaveBytes = 4096;
constexpr uint32_t kExpLdsBytes = kExpWaveOff + kExpWaves * kExpWaveBytes;    // 23440
// the lookback form adds the decoded latents of the chunk's last kExpRing batches (u64[8][256]): a lookback that reaches one of them is
// served from LDS inside the ordered section, one that reaches further back is a global read that can be sent before the batch's turn
constexpr uint32_t kExpRing = 8;
constexpr uint32_t kExpRingOff = kExpLdsBytes;
constexpr uint32_t kExpLbLdsBytes = kExpRingOff + kExpRing * 256 * 8;          // 39824

// What one lane prefetches of one (batch, variable) item: its symbol dword and 32 bytes of the offsets section.
struct ExpPre { uint32_t syms; uint32_t sec[8]; };
constexpr uint32_t kExpStgDwords = 528;   // 2048 B of section + the dwords a 64-bit field may reach into

// Issue the loads of an item.  The section is fetched from its dword-aligned start: lane l takes bytes [32 l, 32 l + 32).
__device__ __forceinline__ void expand_prefetch(ExpPre& pre, gcptr_u8 src, uint64_t src_len, uint64_t start_bit, uint32_t need_bits,
                                                const uint8_t PCO_GLOBAL* syms, uint32_t cnt, bool single_bin) {
  const uint32_t lane = lane_id();
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef u32x4 __attribute__((aligned(1))) u32x4_unaligned;
  pre.syms = (!single_bin && 64 * (lane >> 4) < cnt) ? *(const uint32_t PCO_GLOBAL*)(syms + 4 * lane) : 0u;   // chain (lane / 4) % 4, block 4 (lane / 16) + lane % 4 (see dec_walk_kernel)
  const uint64_t byte0 = (start_bit >> 5) * 4;
  const uint32_t need_bytes = need_bits == 0 ? 0u : (((uint32_t)(start_bit & 31) + need_bits + 31u) / 32u) * 4u + 8u;
  const uint32_t off = 32 * lane;
#pragma unroll
  for (int k = 0; k < 8; k++) pre.sec[k] = 0;
  if (off < need_bytes) {
    if (byte0 + off + 32 <= src_len + 16) {
      const u32x4 a = *(const u32x4_unaligned PCO_GLOBAL*)(src + byte0 + off), b = *(const u32x4_unaligned PCO_GLOBAL*)(src + byte0 + off + 16);
      pre.sec[0] = a.x; pre.sec[1] = a.y; pre.sec[2] = a.z; pre.sec[3] = a.w; pre.sec[4] = b.x; pre.sec[5] = b.y; pre.sec[6] = b.z; pre.sec[7] = b.w;
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) { const uint64_t w = load_u64_le_safe(src, byte0 + off + 8 * k, src_len + 16); pre.sec[2 * k] = (uint32_t)w; pre.sec[2 * k + 1] = (uint32_t)(w >> 32); }
    }
  }
}

// Turn a prefetched item into latents: bins from the symbols, offsets from the staged section (page_latent_decompressor.rs:15-44,179-213).
template <class LV>
__device__ __forceinline__ void expand_item(const ExpPre& pre, uint32_t PCO_LDS* stg, gcptr_u8 src, uint64_t src_len, uint64_t start_bit, uint32_t need_bits, uint32_t cnt,
                                            const uint64_t PCO_LDS* lowers, const uint8_t PCO_LDS* obs, bool single_bin, uint32_t max_ob, LV out[4]) {
  const uint32_t lane = lane_id();
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  if (need_bits != 0) {
    const uint32_t need_bytes = (((uint32_t)(start_bit & 31) + need_bits + 31u) / 32u) * 4u + 8u;
    if (32 * lane < need_bytes) {
      u32x4 a, b; a.x = pre.sec[0]; a.y = pre.sec[1]; a.z = pre.sec[2]; a.w = pre.sec[3]; b.x = pre.sec[4]; b.y = pre.sec[5]; b.z = pre.sec[6]; b.w = pre.sec[7];
      *(u32x4 PCO_LDS*)(stg + 8 * lane) = a; *(u32x4 PCO_LDS*)(stg + 8 * lane + 4) = b;
    }
    if (need_bytes > 2048 && lane < 2) {   // only 64-bit offsets at the maximum width reach past 64 x 32 bytes: fetched here, not prefetched
      const uint64_t w = load_u64_le_safe(src, (start_bit >> 5) * 4 + 2048 + 8 * lane, src_len + 16);
      stg[512 + 2 * lane] = (uint32_t)w; stg[513 + 2 * lane] = (uint32_t)(w >> 32);
    }
  }
  wave_sync_lds();
  // the walker's layout has chain c, block b of a 64-symbol group at dword 4 c + b; this lane wants chain lane % 4 of block lane / 4
  const uint32_t mine = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(4u * ((lane & 48u) + 4u * (lane & 3u) + ((lane >> 2) & 3u))), (int)pre.syms);
  const uint32_t syms = single_bin ? 0u : quad_transpose_u8(mine, lane & 3);
  uint32_t ob[4]; LV low[4]; uint32_t t = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const bool act = 4 * lane + k < cnt;
    const uint32_t s = (syms >> (8 * k)) & 0xffu;
#ifdef PCO_EXP_NOLOOKUP   // (ablation builds only: no LDS bin lookups)
    ob[k] = act ? 8u + (s & 1u) : 0u; low[k] = (LV)s;
#else
    ob[k] = act ? (uint32_t)obs[s] : 0u;
    low[k] = act ? (LV)lowers[s] : (LV)0;
#endif
    t += ob[k];
  }
  const uint32_t incl = wave_incl_scan(t);
  uint32_t r = (uint32_t)(start_bit & 31) + incl - t;
  if (need_bits != 0 && max_ob <= 16) {   // the lane's four fields span at most 64 bits: one window fetch (three dwords), fields cut from registers
    const uint32_t d = r >> 5;
    const uint32_t w0 = stg[d], w1 = stg[d + 1], w2 = stg[d + 2];
    uint64_t v64 = (uint64_t)__builtin_amdgcn_alignbit(w1, w0, r) | ((uint64_t)__builtin_amdgcn_alignbit(w2, w1, r) << 32);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      out[k] = (LV)(low[k] + (LV)__builtin_amdgcn_ubfe((uint32_t)v64, 0u, ob[k]));
      v64 >>= ob[k];
    }
    wave_sync_lds();   // the staging area is reused by the next item
    return;
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {
    LV val = 0;
    if (need_bits != 0) {
      const uint32_t d = r >> 5;
      const uint32_t w0 = stg[d], w1 = stg[d + 1];
      if constexpr (sizeof(LV) == 8) {
        const uint32_t w2 = stg[d + 2];
        const uint64_t v64 = (uint64_t)__builtin_amdgcn_alignbit(w1, w0, r) | ((uint64_t)__builtin_amdgcn_alignbit(w2, w1, r) << 32);
        val = ob[k] >= 64 ? v64 : (v64 & (((uint64_t)1 << ob[k]) - 1));
      } else {
        val = (LV)__builtin_amdgcn_ubfe(__builtin_amdgcn_alignbit(w1, w0, r), 0u, ob[k]);   // offset_bits <= 32; width 32 keeps all bits
        if (ob[k] >= 32) val = (LV)__builtin_amdgcn_alignbit(w1, w0, r);
      }
    }
    out[k] = (LV)(low[k] + val);
    r += ob[k];
  }
  wave_sync_lds();   // the staging area is reused by the next item
}

// One workgroup of kExpWaves waves per chunk.  Unpacking a batch (symbols -> bins -> offsets) is independent of
// every other batch and hides its HBM latency behind the other waves; only the delta decode is ordered: wave b % 4
// enters it when `turn` reaches b, reads the moments the previous batch left in LDS, and passes the turn on.
// kLb: the chunks with a lookback delta (and only those); the plain form leaves them alone.
template <class L, bool kLb>
__global__ __launch_bounds__(256) void dec_expand_kernel(const PcoGfxDecodeTask* tasks, PcoGfxTaskResult* results, const uint32_t* task_ids, uint32_t n_ids,
                                                         const DecPlan* plans, const uint8_t* bins_area, const uint8_t* sym_area, uint64_t sym_stride,
                                                         const uint64_t* offpos_area, uint64_t offpos_stride,
                                                         const uint32_t* progress /* the trailing expanders' done marks, or null */, uint32_t* givebacks) {
  const uint32_t lane = lane_id(), tid = threadIdx.x, wave = tid >> 6;
  uint8_t PCO_LDS* smem = lds_base();
  for (uint32_t bi = blockIdx.x; bi < n_ids; bi += gridDim.x) {
    const uint32_t ti = task_ids ? task_ids[bi] : bi;
    const DecPlan PCO_GLOBAL* plan = (const DecPlan PCO_GLOBAL*)plans + ti;
    const uint32_t pstatus = uni(plan->status);
    if (pstatus == kStatusRetryLegacy) continue;   // the single-kernel decoder finishes this task
    if (uni(plan->fused)) {   // marked for the trailing expanders by the publishing walker (which also wrote its result): skipped only if one of them
                              // expanded it to the end -- a chunk nobody finished (late walker, timed-out or refusing expander) is expanded here
      const bool done = progress != nullptr && uni(progress[(uint64_t)(bi >> 3) * kTrailProgressStride + kTrailDoneWord + (bi & 7u)]) != 0;
      if (!kLb && tid == 0 && givebacks != nullptr) atomicAdd(givebacks + 1, 1u);   // (marked for the expanders)
      if (done) continue;
      if (!kLb && tid == 0 && givebacks != nullptr && pstatus == PCO_GFX_OK) atomicAdd(givebacks, 1u);
    }
    if ((uni(plan->delta_kind[1]) == kDeltaLookback) != kLb) continue;   // the other form's
    if (pstatus != PCO_GFX_OK) {
      if (tid == 0) { PcoGfxTaskResult r; r.n_out = 0; r.consumed = plan->consumed; r.status = pstatus; r.aux = 0; results[ti] = r; }
      continue;
    }
    const PcoGfxDecodeTask task = tasks[ti];
    gcptr_u8 src = (gcptr_u8)task.src;
    const uint64_t src_len = uni((uint64_t)task.src_len);
    L PCO_GLOBAL* dst = (L PCO_GLOBAL*)task.dst;
    const uint32_t n = uni(plan->n), mode_kind = uni(plan->mode_kind), mode_k = uni(plan->mode_k), num_kind = uni(plan->num_kind);
    const L mode_base = (L)uni((uint64_t)plan->mode_base);
    uint32_t present[3], n_bins[3], max_ob[3], dk[3], dord[3], nlps[3];
#pragma unroll
    for (int v = 0; v < 3; v++) {
      present[v] = uni(plan->present[v]); n_bins[v] = uni(plan->n_bins[v]); max_ob[v] = uni(plan->max_ob[v]);
      dk[v] = uni(plan->delta_kind[v]); dord[v] = uni(plan->delta_order[v]); nlps[v] = uni(plan->nlps[v]);
    }
    const uint32_t window_n_log = uni(plan->window_n_log);
    const uint32_t state_n = dk[1] == kDeltaLookback ? nlps[1] : 0u;
    uint64_t PCO_LDS* lowers = (uint64_t PCO_LDS*)(smem + kExpLowOff);
    uint8_t PCO_LDS* obs = smem + kExpObOff;
    uint64_t PCO_LDS* mom64 = (uint64_t PCO_LDS*)(smem + kExpMomOff);
    uint32_t PCO_LDS* turn = (uint32_t PCO_LDS*)(smem + kExpTurnOff);
    uint8_t PCO_LDS* wsm = smem + kExpWaveOff + wave * kExpWaveBytes;
    uint32_t PCO_LDS* dlat = (uint32_t PCO_LDS*)wsm;
    L PCO_LDS* scratch = (L PCO_LDS*)(wsm + 1024);
    uint32_t PCO_LDS* parent = (uint32_t PCO_LDS*)(wsm + 3072);
    uint32_t PCO_LDS* stg = (uint32_t PCO_LDS*)(wsm + 1024);   // section staging; shares scratch / parent, which only the ordered lookback part uses
    L PCO_LDS* moments0 = (L PCO_LDS*)mom64; L PCO_LDS* moments1 = (L PCO_LDS*)(mom64 + 8);
    L PCO_LDS* hring = (L PCO_LDS*)(smem + kExpRingOff);   // (kLb) decoded latents of batch b at [(b % kExpRing) * 256, + 256)
    __syncthreads();   // the previous chunk's LDS state is dead
    const uint8_t PCO_GLOBAL* bins = (const uint8_t PCO_GLOBAL*)bins_area + (uint64_t)ti * kBinsAreaPerTask;
#pragma unroll
    for (int v = 0; v < 3; v++) {
      if (!present[v]) continue;
      const uint64_t PCO_GLOBAL* gl = (const uint64_t PCO_GLOBAL*)(bins + (uint64_t)v * kBinsAreaPerVar);
      const uint8_t PCO_GLOBAL* go = bins + (uint64_t)v * kBinsAreaPerVar + kFastMaxBins * 8;
      for (uint32_t b = tid; b < n_bins[v]; b += 256) { lowers[v * 256 + b] = gl[b]; obs[v * 256 + b] = go[b]; }
    }
    if (tid < 8) { moments0[tid] = (L)plan->moments[0][tid]; moments1[tid] = (L)plan->moments[1][tid]; }
    if (tid == 0) { turn[0] = 0; turn[1] = 0; }
    __syncthreads();
    const bool ordered = dk[1] != kDeltaNone || (present[2] && dk[2] == kDeltaConsecutive);
    const uint32_t n_batches = (n + kBatchN - 1) / kBatchN;
    uint32_t lb_oob = 0;
    // per-variable latent count of a batch
    auto cnt_of = [&](uint32_t batch, int v) -> uint32_t {
      const uint32_t n_remaining = n - batch * kBatchN;
      const uint32_t batch_n = n_remaining < kBatchN ? n_remaining : kBatchN;
      if (v == 0) { const uint32_t lim = n_remaining > nlps[1] ? n_remaining - nlps[1] : 0; return lim < batch_n ? lim : batch_n; }
      const uint32_t rem = n_remaining > nlps[v] ? n_remaining - nlps[v] : 0; return rem < kBatchN ? rem : kBatchN;
    };
    // software pipeline over this wave's batches: the loads of the next batch (symbols, offset sections) are in flight
    // while the current one is expanded; section starts are fetched one batch further ahead.  At most two variables
    // are live on this path (lookback implies classic mode, see fast_front): slot 0 = variable 0 or 2, slot 1 = the primary.
    const int other = present[0] ? 0 : 2;
    const bool has_other = present[0] || present[2];
    const uint32_t wave_u = uni(wave);
    ExpPre pre[2], nxt[2];
    uint64_t st_cur[2] = {0, 0}, st_nxt[2] = {0, 0}, st_raw[2] = {0, 0};
    auto var_of = [&](int slot) -> int { return slot == 1 ? 1 : other; };
    auto issue_starts = [&](uint32_t batch, uint64_t (&st)[2]) {   // plain per-lane loads; made uniform only when they are needed
#pragma unroll
      for (int sl = 0; sl < 2; sl++) {
        const int v = var_of(sl);
        st[sl] = ((sl == 1 || has_other) && batch < n_batches && cnt_of(batch, v) > 0) ? offpos_area[((uint64_t)ti * 3 + v) * offpos_stride + batch] : 0ull;
      }
    };
    auto prefetch = [&](uint32_t batch, const uint64_t (&st)[2], ExpPre (&dstp)[2]) {
#pragma unroll
      for (int sl = 0; sl < 2; sl++) {
        const int v = var_of(sl);
        if (sl == 0 && !has_other) continue;
        const uint32_t cnt = cnt_of(batch, v);
        if (cnt == 0) continue;
        const uint8_t PCO_GLOBAL* syms = (const uint8_t PCO_GLOBAL*)sym_area + ((uint64_t)ti * 3 + v) * sym_stride + (uint64_t)batch * kBatchN;
        expand_prefetch(dstp[sl], src, src_len, st[sl], cnt * max_ob[v], syms, cnt, n_bins[v] <= 1);
      }
    };
    if (wave_u < n_batches) {
      issue_starts(wave_u, st_cur); issue_starts(wave_u + kExpWaves, st_nxt);
      for (int sl = 0; sl < 2; sl++) { st_cur[sl] = uni(st_cur[sl]); st_nxt[sl] = uni(st_nxt[sl]); }
      prefetch(wave_u, st_cur, pre);
    }
    for (uint32_t batch = wave_u; batch < n_batches; batch += kExpWaves) {
      const uint32_t j0 = batch * kBatchN, n_remaining = n - j0;
      const uint32_t batch_n = n_remaining < kBatchN ? n_remaining : kBatchN;
      L prim[4] = {0, 0, 0, 0}, sec[4] = {0, 0, 0, 0};
      uint32_t prim_cnt = 0;
      issue_starts(batch + 2 * kExpWaves, st_raw);   // issued before the prefetch so that waiting for them does not wait for it
      if (batch + kExpWaves < n_batches) prefetch(batch + kExpWaves, st_nxt, nxt);
      // ---- unordered: unpack every variable of the batch ----
#pragma unroll
      for (int sl = 0; sl < 2; sl++) {
        if (sl == 0 && !has_other) continue;
        const int v = var_of(sl);
        const uint32_t cnt = cnt_of(batch, v);
        if (cnt == 0) continue;
        const bool single_bin = n_bins[v] <= 1;
        if (v == 0) {
          uint32_t tmp[4];
          if (max_ob[0] != 0 || !single_bin) expand_item<uint32_t>(pre[0], stg, src, src_len, st_cur[0], cnt * max_ob[0], cnt, lowers, obs, single_bin, max_ob[0], tmp);
          else { const uint32_t l0 = (uint32_t)lowers[0]; for (int k = 0; k < 4; k++) tmp[k] = l0; }
          for (int k = 0; k < 4; k++) dlat[4 * lane + k] = 4 * lane + k < cnt ? tmp[k] : 0u;
        } else {
          L tmp[4];
          if (max_ob[v] != 0 || !single_bin) expand_item<L>(pre[sl], stg, src, src_len, st_cur[sl], cnt * max_ob[v], cnt, lowers + v * 256, obs + v * 256, single_bin, max_ob[v], tmp);
          else { const L l0 = (L)lowers[v * 256]; for (int k = 0; k < 4; k++) tmp[k] = 4 * lane + k < cnt ? l0 : (L)0; }
          if (v == 1) { for (int k = 0; k < 4; k++) prim[k] = tmp[k]; prim_cnt = cnt; } else { for (int k = 0; k < 4; k++) sec[k] = tmp[k]; }
        }
      }
#pragma unroll
      for (int sl = 0; sl < 2; sl++) { pre[sl] = nxt[sl]; st_cur[sl] = st_nxt[sl]; st_nxt[sl] = uni(st_raw[sl]); }
      // (kLb) every element's lookback is known now.  A parent beyond the ring -- more than kExpRing - 1 batches back, or in the page's
      // stored state -- was written at least two laps of the block's waves ago and acknowledged (each wave drains its stores after it
      // passes the turn on): those reads go out here, before the batch waits for its turn.
      L far_val[4] = {0, 0, 0, 0};
      if constexpr (kLb) {
        wave_sync_lds();
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const uint32_t i = 4 * lane + k;
          const uint32_t lb = i < prim_cnt ? dlat[i] : 0u;
          const int64_t q = (int64_t)(state_n + j0 + i) - (int64_t)lb;
          const bool far = lb > i && lb <= (1u << window_n_log) && q >= 0 && (q < (int64_t)state_n || batch - (uint32_t)((q - state_n) >> 8) >= kExpRing);
          if (far) far_val[k] = to_latent_ordered<L>(__hip_atomic_load(&dst[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), num_kind);
        }
      }
      // ---- ordered: delta decode, batch after batch ----
      if (ordered) {
#ifndef PCO_EXP_NOTURN   // (ablation builds only: timing without the batch-to-batch chain; the output is garbage)
        while (__hip_atomic_load((uint32_t*)turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != batch) __builtin_amdgcn_s_sleep(1);
#endif
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (dk[1] == kDeltaConsecutive) consecutive_decode<L>(prim, dord[1], moments0);
        if (present[2] && dk[2] == kDeltaConsecutive) consecutive_decode<L>(sec, dord[2], moments1);
        if constexpr (kLb) {
          // F[state_n + k] = delta_k + MID + F[state_n + k - lb_k] (delta/lookback.rs:200-246).  Parents inside the batch: pointer
          // jumping; in one of the last kExpRing - 1 batches: the ring (no global access inside the ordered section); further back: far_val.
          __builtin_amdgcn_wave_barrier();
          const uint32_t window_n = 1u << window_n_log;
          const uint64_t kbase = (uint64_t)j0;
          for (int k = 0; k < 4; k++) {
            const uint32_t i = 4 * lane + k;
            L val = (L)(prim[k] + lmid<L>());
            uint32_t par = 0xffffffffu;
            if (i < prim_cnt) {
              uint32_t lb = dlat[i];
              if (lb > window_n) { lb_oob = 1; lb = 1; val = (L)(prim[k] + lmid<L>()); }
              if (lb == 0) { }
              else if (lb <= i) par = i - lb;
              else {
                const int64_t q = (int64_t)(state_n + kbase + i) - (int64_t)lb;
                if (q >= 0) {
                  const bool ring_hit = q >= (int64_t)state_n && batch - (uint32_t)((q - state_n) >> 8) < kExpRing;
                  if (ring_hit) val = (L)(val + hring[(uint32_t)(q - state_n) & (kExpRing * 256 - 1)]);
                  else if (dlat[i] <= window_n) val = (L)(val + far_val[k]);
                  else val = (L)(val + to_latent_ordered<L>(__hip_atomic_load(&dst[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), num_kind));   // (an out-of-window lookback, replaced by 1: the page is corrupt and will say so)
                }
              }
            }
            scratch[i] = val; parent[i] = par;
          }
          wave_sync_lds();
          for (int round = 0; round < 8; round++) {   // pointer jumping over the in-batch parents; seasonal data (lookbacks beyond a batch) has none
            L nv[4]; uint32_t np[4]; bool open = false;
            for (int k = 0; k < 4; k++) {
              const uint32_t i = 4 * lane + k; const uint32_t p = parent[i];
              nv[k] = scratch[i]; np[k] = p;
              if (p != 0xffffffffu) { nv[k] = (L)(nv[k] + scratch[p]); np[k] = parent[p]; open = true; }
            }
            if (!__any(open)) break;   // (nothing was read that a lane is about to overwrite: no lane writes in this round)
            wave_sync_lds();
            for (int k = 0; k < 4; k++) { const uint32_t i = 4 * lane + k; scratch[i] = nv[k]; parent[i] = np[k]; }
            wave_sync_lds();
          }
          for (int k = 0; k < 4; k++) {
            const uint32_t i = 4 * lane + k;
            if (i < prim_cnt) {
              const L f = scratch[i];
              hring[(uint32_t)(kbase + i) & (kExpRing * 256 - 1)] = f;
              dst[state_n + kbase + i] = from_latent_ordered<L>(f, num_kind);
            }
          }
          __builtin_amdgcn_wave_barrier();
        }
        else if (dk[1] == kDeltaLookback) {   // (never taken: the plain form skips lookback chunks.  Kept because without it the compiler schedules the 32-bit instantiation's main loop 15 % slower)
          // history lives in dst: earlier batches' stores were acknowledged by L2 before their wave passed the turn on
          // (s_waitcnt vmcnt(0) below), and the loads here are agent-scope atomics, i.e. served by L2
          __builtin_amdgcn_wave_barrier();
          const uint32_t window_n = 1u << window_n_log;
          const uint64_t kbase = (uint64_t)j0;
          for (int k = 0; k < 4; k++) {
            const uint32_t i = 4 * lane + k;
            L val = (L)(prim[k] + lmid<L>());
            uint32_t par = 0xffffffffu;
            if (i < prim_cnt) {
              uint32_t lb = dlat[i];
              if (lb > window_n) { lb_oob = 1; lb = 1; }
              if (lb == 0) { }
              else if (lb <= i) par = i - lb;
              else {
                const int64_t jsrc = (int64_t)(state_n + kbase + i) - (int64_t)lb;
                if (jsrc >= 0) val = (L)(val + to_latent_ordered<L>(__hip_atomic_load(&dst[jsrc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), num_kind));
              }
            }
            scratch[i] = val; parent[i] = par;
          }
          wave_sync_lds();
          for (int round = 0; round < 8; round++) {
            L nv[4]; uint32_t np[4];
            for (int k = 0; k < 4; k++) {
              const uint32_t i = 4 * lane + k; const uint32_t p = parent[i];
              nv[k] = scratch[i]; np[k] = p;
              if (p != 0xffffffffu) { nv[k] = (L)(nv[k] + scratch[p]); np[k] = parent[p]; }
            }
            wave_sync_lds();
            for (int k = 0; k < 4; k++) { const uint32_t i = 4 * lane + k; scratch[i] = nv[k]; parent[i] = np[k]; }
            wave_sync_lds();
          }
          for (int k = 0; k < 4; k++) { const uint32_t i = 4 * lane + k; if (i < prim_cnt) dst[state_n + kbase + i] = from_latent_ordered<L>(scratch[i], num_kind); }
          __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __builtin_amdgcn_wave_barrier();
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) __hip_atomic_store((uint32_t*)turn, batch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if constexpr (kLb) __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");   // off the chunk's serial chain: this batch's numbers are in L2 before this wave reads or sends anything else
      }
      if (!kLb && dk[1] != kDeltaLookback) {
        L outv[4];
#pragma unroll
        for (int k = 0; k < 4; k++) outv[k] = join_one<L>(mode_kind, num_kind, mode_base, mode_k, prim[k], sec[k]);
        const uint32_t i0 = 4 * lane;
        L PCO_GLOBAL* o = dst + j0 + i0;
#ifdef PCO_EXP_NOSTORE   // (ablation builds only: the kernel without its output stream)
        if ((outv[0] ^ outv[1] ^ outv[2] ^ outv[3]) == (L)0x9e3779b97f4a7c15ull)
#endif
        if (sizeof(L) == 8 && batch_n == kBatchN && (((uintptr_t)(dst + j0)) & 15) == 0) {   // (uniform: a full batch to an aligned place)
          if constexpr (sizeof(L) == 8) { const unsigned long long y[4] = {outv[0], outv[1], outv[2], outv[3]}; store_u64_batch((unsigned long long PCO_GLOBAL*)(dst + j0), y); }
        } else if (i0 + 4 <= batch_n && (((uintptr_t)o) & 15) == 0) {
          if constexpr (sizeof(L) == 8) {
            typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
            u64x2 PCO_GLOBAL* p = (u64x2 PCO_GLOBAL*)o;
            u64x2 a; a.x = outv[0]; a.y = outv[1]; u64x2 b; b.x = outv[2]; b.y = outv[3];
            p[0] = a; p[1] = b;
          } else if constexpr (sizeof(L) == 4) {
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            u32x4 a; a.x = outv[0]; a.y = outv[1]; a.z = outv[2]; a.w = outv[3];
            __builtin_nontemporal_store(a, (u32x4 PCO_GLOBAL*)o);   // (a contiguous KB per instruction, and nobody on the device reads the numbers back)
          } else { for (int k = 0; k < 4; k++) o[k] = outv[k]; }
        } else { for (int k = 0; k < 4; k++) if (i0 + k < batch_n) o[k] = outv[k]; }
      }
    }
    if (uni(wave_or_u32(lb_oob)) && lane == 0) __hip_atomic_fetch_or((uint32_t*)(turn + 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    if (tid == 0) {
      const uint32_t status = turn[1] ? PCO_GFX_CORRUPTION : PCO_GFX_OK;
      PcoGfxTaskResult r; r.n_out = status == PCO_GFX_OK ? n : 0; r.consumed = plan->consumed; r.status = status; r.aux = plan->more; results[ti] = r;
    }
  }
}

}  // namespace pcogfx
