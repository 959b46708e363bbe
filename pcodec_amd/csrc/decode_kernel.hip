// decode_kernel.hip -- gfx950 chunk decoder: one standalone chunk stream per wavefront.
//
// Replaces, for the device path, the reference's decode stack
//   standalone/decompressor.rs:85-284 (file header, chunk preamble)
//   metadata/{chunk,chunk_latent_var,mode,delta_encoding,page,page_latent_var}.rs (parsing)
//   chunk_latent_decompressor.rs:30-76 + ans/{spec,decoding}.rs (tANS decoder tables)
//   page_latent_decompressor.rs:15-257 (tANS walk, offset unpack)
//   delta/{consecutive,lookback}.rs decode, mode/*.rs join_latents
//   wrapped/page_decompressor.rs:115-191 (batch driver)
//
// Design (CDNA4): a page is one serial tANS stream (the bit cursor is data dependent), so the
// unit of parallelism is the chunk.  One 64-lane wave owns one chunk: lanes 0-3 walk the four
// interleaved tANS chains with the decoder table in LDS and the batch's ANS bits staged in an
// LDS window; then all 64 lanes unpack the batch's 256 offsets (4 contiguous latents per lane),
// run the delta scan with wave shuffles, join and store 32 contiguous bytes per lane.  All
// metadata is parsed on the device, so a many-chunk decode is a single launch with no host
// round trip.
#pragma once
#include <type_traits>

#include "pco_dev.h"

namespace pcogfx {

#define PCO_LDS __attribute__((address_space(3)))
template <bool kLds, class T> using tptr = std::conditional_t<kLds, T PCO_LDS*, T PCO_GLOBAL*>;

// Where a wrapped page's ChunkMeta lives when it is not in front of the page (pco_gfx_decompress_pages: one ChunkMeta, many pages); indexed
// like the tasks.  p == nullptr: the task's src starts with the ChunkMeta (PCO_GFX_TASK_WRAPPED_PAGE as the host-buffer entry points use it).
struct MetaRef { const void* p; uint64_t len; };

struct VarInfo {
  uint32_t present, latent_bits, ans_size_log, n_bins, max_ob;
  uint32_t delta_kind, delta_order, window_n_log, state_n_log;  // LatentVarDeltaEncoding
  uint32_t off_nodes, off_lower, off_ob, off_cum;               // byte offsets into the table area
  uint32_t pad[3];
};
static_assert(sizeof(VarInfo) == 64, "VarInfo");

constexpr uint32_t kLdsWinOff = 0;                          // u32[136]: one batch's ANS section (<= 448 B) + slack
constexpr uint32_t kLdsSymOff = 544;                        // u16[256]
constexpr uint32_t kLdsMomOff = kLdsSymOff + 512;           // u64[2][8] consecutive-delta moments
constexpr uint32_t kLdsVarOff = kLdsMomOff + 128;           // VarInfo[3]
constexpr uint32_t kLdsDlatOff = kLdsVarOff + 192;          // u32[256] lookback latents of the batch
constexpr uint32_t kLdsScratchOff = kLdsDlatOff + 1024;     // u64[256] lookback pointer jumping values
constexpr uint32_t kLdsParentOff = kLdsScratchOff + 2048;   // u32[256] lookback parents
constexpr uint32_t kLdsFixed = kLdsParentOff + 1024;        // 5472; tANS tables follow
static_assert(kLdsFixed % 16 == 0, "LDS carve must stay 16-byte aligned");
// per-wave global table workspace (used only when a chunk's tables exceed the LDS budget)
constexpr uint64_t kTblWsBytesPerVar = (4ull << kMaxAnsBits) + (1ull << kMaxAnsBits) * (8 + 1 + 4) + 256;
constexpr uint64_t kTblWsBytes = 3 * kTblWsBytesPerVar;

__device__ __forceinline__ uint8_t PCO_LDS* lds_base() {
  extern __shared__ __attribute__((aligned(16))) uint8_t pco_lds[];
  return (uint8_t PCO_LDS*)pco_lds;
}
__device__ __forceinline__ void wave_sync_lds() {
  // one wave per workgroup: LDS ops of a wave execute in order; this only stops the compiler reordering them
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint32_t var_table_bytes(uint32_t ans_size_log, uint32_t n_bins, uint32_t latent_bytes) {
  uint32_t b = (4u << ans_size_log);
  b += ((n_bins * latent_bytes + 7u) & ~7u);
  b += ((n_bins + 7u) & ~7u);
  b += (((n_bins + 1) * 4u + 7u) & ~7u);
  return b;
}

// node: bits 0..13 next_state_idx_base, 14..17 bits_to_read, 18..31 symbol
__device__ __forceinline__ uint32_t make_node(uint32_t base, uint32_t btr, uint32_t sym) { return base | (btr << 14) | (sym << 18); }

// Build one variable's decoder tables (ans/spec.rs:37-59, ans/decoding.rs:27-47).  Returns false on error.
template <class LV, bool kLds>
__device__ __noinline__ bool build_var_tables(tptr<kLds, uint8_t> tbl, uint32_t vi, MetaReader& mr, uint32_t& status) {
  const uint32_t lane = lane_id();
  VarInfo PCO_LDS* vinfo = (VarInfo PCO_LDS*)(lds_base() + kLdsVarOff) + vi;
  const uint32_t asl = uni(vinfo->ans_size_log), n_bins = uni(vinfo->n_bins), latent_bits = uni(vinfo->latent_bits);
  const uint32_t T = 1u << asl;
  tptr<kLds, uint32_t> nodes = (tptr<kLds, uint32_t>)(tbl + uni(vinfo->off_nodes));
  tptr<kLds, LV> lowers = (tptr<kLds, LV>)(tbl + uni(vinfo->off_lower));
  tptr<kLds, uint8_t> obs = tbl + uni(vinfo->off_ob);
  tptr<kLds, uint32_t> cum = (tptr<kLds, uint32_t>)(tbl + uni(vinfo->off_cum));
  const uint32_t obb = offset_bits_bits(latent_bits);
  const uint32_t bin_bits = asl + latent_bits + obb;
  const uint64_t bins_start = mr.bit;
  // parse bins in parallel (metadata/chunk_latent_var.rs:21-53)
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
      lowers[b] = (LV)lower; obs[b] = (uint8_t)ob;
      max_ob = max_ob > ob ? max_ob : ob;
    }
    const uint32_t incl = wave_incl_scan(w);
    if (b < n_bins) cum[b] = carry + incl - w;
    carry += wave_last(incl);
  }
  if (lane == 0) { cum[n_bins] = carry; }
  mr.bit = bins_start + (uint64_t)n_bins * bin_bits;
  max_ob = uni(wave_max_u32(max_ob));
  if (lane == 0) vinfo->max_ob = max_ob;
  if (!mr.in_bounds()) { status = PCO_GFX_INSUFFICIENT_DATA; return false; }
  if (uni(wave_or_u32(bad))) { status = PCO_GFX_CORRUPTION; return false; }
  if (n_bins == 0) {  // Spec::from_weights: empty -> [1]; a single node, never walked
    if (lane == 0) nodes[0] = make_node(0, 0, 0);
    wave_sync_lds();
    return true;
  }
  if (carry != T) { status = PCO_GFX_CORRUPTION; return false; }  // "table size log does not agree with total weight"
  wave_sync_lds();
  // state symbols: state[(stride*step) & (T-1)] = symbol of step
  uint32_t stride = (3 * T) / 5; if ((stride & 1) == 0) stride += 1;
  for (uint32_t t = lane; t < T; t += 64) {
    uint32_t lo = 0, hi = n_bins;  // invariant cum[lo] <= t < cum[hi]
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (cum[mid] <= t) lo = mid; else hi = mid; }
    nodes[(stride * t) & (T - 1)] = lo;
  }
  wave_sync_lds();
  // x_s = weight + rank among same-symbol states in ascending state order; cum[] becomes the running counter
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
    const uint32_t s = act ? nodes[i] : 0xffffffffu;
    uint64_t m = __ballot(act);
    for (uint32_t bit = 0; bit < sym_bits; bit++) {  // match-any over the symbol bits
      const uint64_t bm = __ballot((s >> bit) & 1);
      m &= ((s >> bit) & 1) ? bm : ~bm;
    }
    const uint64_t lt = ((uint64_t)1 << lane) - 1;
    const uint32_t rank = __popcll(m & lt), gcount = __popcll(m);
    const uint32_t basec = act ? cum[s] : 0;
    wave_sync_lds();
    if (act && rank == 0) cum[s] = basec + gcount;
    wave_sync_lds();
    if (act) {
      const uint32_t x_s = basec + rank;
      const uint32_t btr = clz_u32(x_s) - clz_u32(T);
      nodes[i] = make_node((x_s << btr) - T, btr, s);
    }
  }
  wave_sync_lds();
  return true;
}

// Walk `cnt` tANS symbols of one variable (page_latent_decompressor.rs:89-177).  Lanes 0..3 hold the
// four chain states in `st`.  Symbols land in sym_stage[0..cnt).  Returns the new bit position.
template <bool kLds>
__device__ __forceinline__ uint64_t walk_ans(gcptr_u8 src, uint64_t src_len, uint64_t bitpos, uint32_t cnt,
                                              tptr<kLds, uint32_t> nodes, uint32_t& st) {
  const uint32_t lane = lane_id();
  uint32_t PCO_LDS* win = (uint32_t PCO_LDS*)(lds_base() + kLdsWinOff);
  uint16_t PCO_LDS* sym_stage = (uint16_t PCO_LDS*)(lds_base() + kLdsSymOff);
  // stage the ANS section: 136 dwords from the dword containing bitpos
  const uint64_t dw0 = bitpos >> 5;
  {
    const uint64_t w = load_u64_le_safe(src, dw0 * 4 + (uint64_t)lane * 8, src_len + 16);
    win[2 * lane] = (uint32_t)w; win[2 * lane + 1] = (uint32_t)(w >> 32);
    if (lane < 4) {
      const uint64_t w2 = load_u64_le_safe(src, dw0 * 4 + 512 + (uint64_t)lane * 8, src_len + 16);
      win[128 + 2 * lane] = (uint32_t)w2; win[128 + 2 * lane + 1] = (uint32_t)(w2 >> 32);
    }
  }
  wave_sync_lds();
  uint32_t rel = (uint32_t)(bitpos & 31);  // bit offset relative to win[0]
  if (lane < 4) {
    const uint32_t steps = (cnt + 3) >> 2;
    for (uint32_t g = 0; g < steps; g++) {
      const bool act = 4 * g + lane < cnt;
      const uint32_t node = nodes[st];
      const uint32_t d = rel >> 5;
      const uint32_t w0 = win[d], w1 = win[d + 1], w2 = win[d + 2], w3 = win[d + 3];
      const uint32_t btr = act ? ((node >> 14) & 15u) : 0u;
      // exclusive prefix of btr over the quad (DPP quad_perm)
      uint32_t p1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)btr, 0x90 /*[0,0,1,2]*/, 0xf, 0xf, false);
      p1 = lane >= 1 ? p1 : 0;
      uint32_t incl = btr + p1;
      uint32_t p2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)incl, 0x40 /*[0,0,0,1]*/, 0xf, 0xf, false);
      p2 = lane >= 2 ? p2 : 0;
      incl += p2;
      const uint32_t excl = incl - btr;
      const uint32_t tot = (uint32_t)__builtin_amdgcn_mov_dpp((int)incl, 0xff /*[3,3,3,3]*/, 0xf, 0xf, false);
      const uint32_t sh = (rel & 31) + excl;  // <= 31 + 42
      const uint32_t sel = sh >> 5;
      const uint32_t lo = sel == 0 ? w0 : (sel == 1 ? w1 : w2);
      const uint32_t hi = sel == 0 ? w1 : (sel == 1 ? w2 : w3);
      const uint32_t bits = __builtin_amdgcn_alignbit(hi, lo, sh & 31);
      const uint32_t val = bits & ((1u << btr) - 1u);
      if (act) {
        sym_stage[4 * g + lane] = (uint16_t)(node >> 18);
        st = (node & 0x3fffu) + val;
      }
      rel += tot;
    }
  }
  const uint32_t rel_end = uni(rel);  // lane 0
  wave_sync_lds();
  return (dw0 << 5) + rel_end;
}

// Unpack offsets and add lowers for one variable's batch (page_latent_decompressor.rs:15-44,181-235).
// Lane l produces latents for positions 4l..4l+3 (zero beyond cnt).  Returns the new bit position.
template <class LV, bool kLds>
__device__ __forceinline__ uint64_t unpack_offsets(gcptr_u8 src, uint64_t src_len, uint64_t bitpos, uint32_t cnt,
                                                    tptr<kLds, LV> lowers, tptr<kLds, uint8_t> obs, bool single_bin, LV out[4]) {
  const uint32_t lane = lane_id();
  uint16_t PCO_LDS* sym_stage = (uint16_t PCO_LDS*)(lds_base() + kLdsSymOff);
  uint32_t ob[4]; LV low[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t i = 4 * lane + k;
    const bool act = i < cnt;
    const uint32_t s = (single_bin || !act) ? 0u : (uint32_t)sym_stage[i];
    ob[k] = act ? (uint32_t)obs[s] : 0u;
    low[k] = act ? lowers[s] : (LV)0;
  }
  const uint32_t t = ob[0] + ob[1] + ob[2] + ob[3];
  const uint32_t incl = wave_incl_scan(t);
  const uint32_t total = wave_last(incl);
  uint64_t b = bitpos + (incl - t);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    uint64_t val = 0;
    if (ob[k] != 0) {
      const uint64_t byte = b >> 3; const uint32_t sh = (uint32_t)(b & 7);
      val = load_u64_le_safe(src, byte, src_len + 16) >> sh;
      if (sh + ob[k] > 64) val |= load_u64_le_safe(src, byte + 8, src_len + 16) << (64 - sh);
      if (ob[k] < 64) val &= ((uint64_t)1 << ob[k]) - 1;
    }
    out[k] = (LV)(low[k] + (LV)val);
    b += ob[k];
  }
  return bitpos + total;
}

// Consecutive delta decode of one batch (delta/consecutive.rs:35-50): toggle centre, then `order`
// running sums seeded by the carried moments (highest order first).
template <class L>
__device__ __forceinline__ void consecutive_decode(L x[4], uint32_t order, L PCO_LDS* moments) {
  const uint32_t lane = lane_id();
#pragma unroll
  for (int k = 0; k < 4; k++) x[k] = (L)(x[k] + lmid<L>());
  for (uint32_t m = order; m-- > 0;) {
    const L mom = moments[m];
    const L e1 = x[0], e2 = (L)(e1 + x[1]), e3 = (L)(e2 + x[2]), t = (L)(e3 + x[3]);
    const L incl = wave_incl_scan(t);
    const L base = (L)(mom + (L)(incl - t));
    x[0] = base; x[1] = (L)(base + e1); x[2] = (L)(base + e2); x[3] = (L)(base + e3);
    const L total = wave_last(incl);
    wave_sync_lds();
    if (lane == 0) moments[m] = (L)(mom + total);
    wave_sync_lds();
  }
}

// join (mode/classic.rs:14-24, int_mult.rs:38-54, float_mult.rs:17-36, float_quant.rs:13-39)
template <class L>
__device__ __forceinline__ L join_one(uint32_t mode_kind, uint32_t num_kind, L base_latent, uint32_t k, L p, L s) {
  switch (mode_kind) {
    case kClassic: return from_latent_ordered<L>(p, num_kind);
    case kIntMult: return from_latent_ordered<L>((L)((L)(p * base_latent) + s), num_kind);
    case kFloatQuant: {
      const L sign_cutoff = (L)(lmid<L>() >> k);
      const L lowmax = (L)(((L)1 << k) - 1);
      const L low = p >= sign_cutoff ? s : (L)(lowmax - s);
      return from_latent_ordered<L>((L)((L)(p << k) + low), kFloat);
    }
    default: {  // kFloatMult
      if constexpr (sizeof(L) >= 4) {
        typedef typename FloatOf<L>::F F;
        const F base = bits_to_float(from_latent_ordered<L>(base_latent, kFloat));
        const F unadjusted = int_float_from_latent<L>(p) * base;
        const L l = (L)(to_latent_ordered<L>(float_to_bits(unadjusted), kFloat) + s + lmid<L>());
        return from_latent_ordered<L>(l, kFloat);
      } else if constexpr (sizeof(L) == 2) {   // f16 (pco_dev.h)
        const uint32_t base = (uint32_t)from_latent_ordered<L>(base_latent, kFloat);
        const uint32_t unadjusted = half_mul(half_int_float_from_latent((uint32_t)p), base);
        const L l = (L)(to_latent_ordered<L>((L)unadjusted, kFloat) + s + lmid<L>());
        return from_latent_ordered<L>(l, kFloat);
      } else return 0;
    }
  }
}

struct PageParams {
  uint32_t mode_kind, mode_k, num_kind, n;
  uint64_t mode_base;
  uint64_t dict_byte; uint32_t dict_n;     // Dict mode: the dictionary's first byte in the ChunkMeta's buffer and its length (metadata/mode.rs:138-165)
  uint32_t conv_order, conv_quant;         // Conv1 delta (metadata/delta_encoding.rs): weights and bias live in LDS (kLdsConvOff)
  void PCO_GLOBAL* sec_hist;               // lookback with a delta'd SECONDARY variable: n latents of scratch for its history (else null)
  gcptr_u8 dict_src = nullptr;             // ... and that buffer (the page body may live in a buffer of its own: PcoGfxPageTask)
  uint32_t* progress = nullptr;            // (wrapped pages) where to say how far a page got that then failed: 1 + the numbers of the batches before the failing one, 0 = before its first batch was reached
};
// Conv1 parameters in LDS, in the lookback path's "parent" area (the two deltas exclude each other): i64 bias | i64 weights[32]
constexpr uint32_t kLdsConvOff = kLdsParentOff;

// Conv1 delta decode of one batch (delta/conv1.rs:231-251,463-484): residuals = state ++ (deltas + MID); every residual past the state
// gets max(0, bias + sum_k weights[k] * residuals[i - order + k]) >> quantization added, in the latent's "Conv" integer type (i16 / i32 /
// i64 for 8- / 16- / 32-bit latents: sums formed in 64 bits and wrapped to that width); the batch's numbers are the first dst_n residuals,
// the last `order` become the next state.  A serial recurrence: lane 0 walks it (this is the general kernel; Conv1 is rare).
template <class L>
__device__ __forceinline__ void conv1_decode(L x[4], uint32_t dst_n, uint32_t order, uint32_t quant) {
  const uint32_t lane = lane_id();
  uint32_t PCO_LDS* res = (uint32_t PCO_LDS*)(lds_base() + kLdsScratchOff);   // u32[32 + 256]: state, then this batch
  const int64_t PCO_LDS* cw = (const int64_t PCO_LDS*)(lds_base() + kLdsConvOff);
  constexpr int kConvBits = LBits<L>::v == 32 ? 64 : 2 * (int)LBits<L>::v;
  auto wrap = [&](uint64_t v) -> int64_t { return kConvBits == 64 ? (int64_t)v : (int64_t)(v << (64 - kConvBits)) >> (64 - kConvBits); };
#pragma unroll
  for (int k = 0; k < 4; k++) res[order + 4 * lane + k] = (uint32_t)(L)(x[k] + lmid<L>());
  wave_sync_lds();
  if (lane == 0) {
    const int64_t bias = wrap((uint64_t)cw[0]);
    for (uint32_t i = 0; i < dst_n; i++) {
      uint64_t sum = (uint64_t)bias;
      for (uint32_t k = 0; k < order; k++) sum += (uint64_t)wrap((uint64_t)cw[1 + k]) * (uint64_t)res[i + k];
      int64_t sc = wrap(sum);
      if (sc < 0) sc = 0;
      res[order + i] = (uint32_t)(L)((L)res[order + i] + (L)(uint64_t)(sc >> quant));
    }
  }
  wave_sync_lds();
#pragma unroll
  for (int k = 0; k < 4; k++) x[k] = (L)res[4 * lane + k];
  wave_sync_lds();
  if (lane == 0) for (uint32_t j = 0; j < order; j++) res[j] = res[dst_n + j];   // (ascending: every source index is above its target)
  wave_sync_lds();
}

// Decode one page (page meta + all batches) with number latent type L.
template <class L, bool kLds>
__device__ __noinline__ void decode_page_body(gcptr_u8 src, uint64_t src_len, MetaReader& mr, tptr<kLds, uint8_t> tbl,
                                             const PageParams& pp, L PCO_GLOBAL* dst, uint32_t& status) {
  const uint32_t lane = lane_id();
  uint8_t PCO_LDS* smem = lds_base();
  uint64_t PCO_LDS* mom64 = (uint64_t PCO_LDS*)(smem + kLdsMomOff);
  uint32_t PCO_LDS* dlat = (uint32_t PCO_LDS*)(smem + kLdsDlatOff);
  L PCO_LDS* scratch = (L PCO_LDS*)(smem + kLdsScratchOff);
  uint32_t PCO_LDS* parent = (uint32_t PCO_LDS*)(smem + kLdsParentOff);
  VarInfo PCO_LDS* vinfo = (VarInfo PCO_LDS*)(smem + kLdsVarOff);
  L PCO_LDS* moments0 = (L PCO_LDS*)(mom64);
  L PCO_LDS* moments1 = (L PCO_LDS*)(mom64 + 8);
  const uint32_t mode_kind = pp.mode_kind, mode_k = pp.mode_k, num_kind = pp.num_kind, n = pp.n;
  const L mode_base = (L)pp.mode_base;

  // wave-uniform per-variable parameters (SGPRs)
  uint32_t present[3], n_bins[3], max_ob[3], asl[3], dk[3], dord[3], nlps[3], off_nodes[3], off_lower[3], off_ob[3];
#pragma unroll
  for (int vi = 0; vi < 3; vi++) {
    present[vi] = uni(vinfo[vi].present); n_bins[vi] = uni(vinfo[vi].n_bins); max_ob[vi] = uni(vinfo[vi].max_ob);
    asl[vi] = uni(vinfo[vi].ans_size_log); dk[vi] = uni(vinfo[vi].delta_kind); dord[vi] = uni(vinfo[vi].delta_order);
    off_nodes[vi] = uni(vinfo[vi].off_nodes); off_lower[vi] = uni(vinfo[vi].off_lower); off_ob[vi] = uni(vinfo[vi].off_ob);
    nlps[vi] = (dk[vi] == kDeltaConsecutive || dk[vi] == kDeltaConv1) ? dord[vi] : (dk[vi] == kDeltaLookback ? (1u << uni(vinfo[vi].state_n_log)) : 0u);
  }
  const uint32_t window_n_log = uni(vinfo[1].window_n_log);
  const uint32_t state_n = dk[1] == kDeltaLookback ? nlps[1] : 0u;
  // Lookback keeps its history in dst.  In classic mode dst holds the numbers, whose ordered latents ARE the primary latents; in the
  // other modes a first pass leaves the primary latents themselves in dst, and a second pass over the page decodes the secondary
  // variable again and joins in place (this combination only comes from explicit specs or Auto on unusual data; it is not fast).
  const bool raw_hist = dk[1] == kDeltaLookback && mode_kind != kClassic;
  const uint32_t n_pass = raw_hist ? 2u : 1u;

  // ---- page meta (metadata/page.rs:36-57, page_latent_var.rs:28-49) ----
  uint32_t st[3] = {0, 0, 0};
#pragma unroll
  for (int vi = 0; vi < 3; vi++) {
    if (!present[vi]) continue;
    const uint32_t lbits = vi == 0 ? 32u : LBits<L>::v;
    for (uint32_t i = 0; i < nlps[vi]; i++) {
      const L x = (L)mr.read(lbits);
      if (dk[vi] == kDeltaConsecutive) { if (lane == 0) { if (vi == 2) moments1[i] = x; else moments0[i] = x; } }
      else if (dk[vi] == kDeltaConv1) { if (lane == 0 && i < 32) ((uint32_t PCO_LDS*)(smem + kLdsScratchOff))[i] = (uint32_t)x; }
      else if (vi == 1 && i < n && mr.in_bounds()) {  // lookback state = the first state_n latents
        if (lane == 0) dst[i] = raw_hist ? x : from_latent_ordered<L>(x, num_kind);
      }
      else if (vi == 2 && dk[2] == kDeltaLookback && i < n && mr.in_bounds()) { if (lane == 0) ((L PCO_GLOBAL*)pp.sec_hist)[i] = x; }
    }
    uint32_t mine = 0;
    for (uint32_t j = 0; j < 4; j++) { const uint32_t s = (uint32_t)mr.read(asl[vi]); if (lane == j) mine = s; }
    st[vi] = mine;
  }
  if (!mr.drain_empty_byte()) { if (mr.in_bounds()) status = PCO_GFX_CORRUPTION; }
  if (!mr.in_bounds()) { status = PCO_GFX_INSUFFICIENT_DATA; return; }
  if (status) return;
  const uint32_t n_in_body = n > nlps[1] ? n - nlps[1] : 0;
  if (n_in_body > 0) {
#pragma unroll
    for (int vi = 0; vi < 3; vi++) if (present[vi] && n_bins[vi] == 0) { status = PCO_GFX_CORRUPTION; return; }
  }
  wave_sync_lds();

  uint64_t bitpos = mr.bit;
  const uint64_t body_start = mr.bit;
  const uint32_t st_init[3] = {st[0], st[1], st[2]};
  uint32_t n_remaining = n;
  uint32_t lb_oob = 0;
  for (uint32_t pass = 0; pass < n_pass; pass++) {
  if (pass == 1) {
    if (uni(wave_or_u32(lb_oob))) { status = PCO_GFX_CORRUPTION; return; }
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bitpos = body_start; n_remaining = n; st[0] = st_init[0]; st[1] = st_init[1]; st[2] = st_init[2];
  }
  for (uint32_t j0 = 0; j0 < n; j0 += kBatchN) {
    const uint32_t batch_n = n_remaining < kBatchN ? n_remaining : kBatchN;
    L prim[4] = {0, 0, 0, 0}, sec[4] = {0, 0, 0, 0};
    uint32_t prim_cnt = 0;
#pragma unroll
    for (int vi = 0; vi < 3; vi++) {
      if (!present[vi]) continue;
      uint32_t cnt;
      if (vi == 0) { const uint32_t lim = n_remaining > nlps[1] ? n_remaining - nlps[1] : 0; cnt = lim < batch_n ? lim : batch_n; }
      else { const uint32_t rem = n_remaining > nlps[vi] ? n_remaining - nlps[vi] : 0; cnt = rem < kBatchN ? rem : kBatchN; }
      if (cnt > 0) {
        const bool single_bin = n_bins[vi] <= 1;
        if (!single_bin) bitpos = walk_ans<kLds>(src, src_len, bitpos, cnt, (tptr<kLds, uint32_t>)(tbl + off_nodes[vi]), st[vi]);
        if (vi == 0) {
          uint32_t tmp[4];
          tptr<kLds, uint32_t> lw = (tptr<kLds, uint32_t>)(tbl + off_lower[0]);
          if (max_ob[0] != 0 || !single_bin) bitpos = unpack_offsets<uint32_t, kLds>(src, src_len, bitpos, cnt, lw, tbl + off_ob[0], single_bin, tmp);
          else { const uint32_t l0 = lw[0]; for (int k = 0; k < 4; k++) tmp[k] = l0; }
          for (int k = 0; k < 4; k++) dlat[4 * lane + k] = 4 * lane + k < cnt ? tmp[k] : 0u;
        } else {
          L tmp[4];
          tptr<kLds, L> lw = (tptr<kLds, L>)(tbl + off_lower[vi]);
          if (max_ob[vi] != 0 || !single_bin) bitpos = unpack_offsets<L, kLds>(src, src_len, bitpos, cnt, lw, tbl + off_ob[vi], single_bin, tmp);
          else { const L l0 = lw[0]; for (int k = 0; k < 4; k++) tmp[k] = 4 * lane + k < cnt ? l0 : (L)0; }
          if (vi == 1) { for (int k = 0; k < 4; k++) prim[k] = tmp[k]; prim_cnt = cnt; } else { for (int k = 0; k < 4; k++) sec[k] = tmp[k]; }
        }
        if (bitpos > src_len * 8) { status = PCO_GFX_INSUFFICIENT_DATA; if (pp.progress && n_pass == 1) *pp.progress = 1u + j0; return; }
      }
      // delta decode (delta/mod.rs:125-159)
      if (vi >= 1 && dk[vi] == kDeltaConsecutive) {
        if (vi == 1) consecutive_decode<L>(prim, dord[1], moments0); else consecutive_decode<L>(sec, dord[2], moments1);
      }
      else if (vi == 1 && dk[1] == kDeltaConv1) {
        if constexpr (sizeof(L) <= 4) conv1_decode<L>(prim, batch_n, dord[1], pp.conv_quant);
      }
    }
    if (dk[1] == kDeltaLookback && pass == 0) {
      // F[state_n + k] = delta_k + MID + F[state_n + k - lb_k]; history lives in dst (numbers in classic mode, else primary latents).
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      __builtin_amdgcn_wave_barrier();
      const uint32_t window_n = 1u << window_n_log;
      const uint64_t kbase = (uint64_t)j0;  // index of this batch's first delta
      for (int k = 0; k < 4; k++) {
        const uint32_t i = 4 * lane + k;
        L val = (L)(prim[k] + lmid<L>());
        uint32_t par = 0xffffffffu;
        if (i < prim_cnt) {
          uint32_t lb = dlat[i];
          if (lb > window_n) { lb_oob = 1; lb = 1; }
          if (lb == 0) {
            // the reference adds the slot's stale window content here; only reachable from corrupt bins
          } else if (lb <= i) par = i - lb;
          else {
            const int64_t jsrc = (int64_t)(state_n + kbase + i) - (int64_t)lb;
            if (jsrc >= 0) { const L h = __hip_atomic_load(&dst[jsrc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); val = (L)(val + (raw_hist ? h : to_latent_ordered<L>(h, num_kind))); }
          }
        }
        scratch[i] = val; parent[i] = par;
      }
      wave_sync_lds();
      for (int round = 0; round < 8; round++) {  // pointer jumping: resolves in-batch chains of length <= 256
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
      for (int k = 0; k < 4; k++) {  // store F at its true index (the reference's output lags by state_n)
        const uint32_t i = 4 * lane + k;
        if (i < prim_cnt) dst[state_n + kbase + i] = raw_hist ? scratch[i] : from_latent_ordered<L>(scratch[i], num_kind);
      }
      if (dk[2] == kDeltaLookback) {
        // the secondary variable follows the same lookbacks (delta/mod.rs:125-159: both variables are decoded against the one delta
        // variable); its history lives in the task's scratch
        L PCO_GLOBAL* sh = (L PCO_GLOBAL*)pp.sec_hist;
        wave_sync_lds();
        for (int k = 0; k < 4; k++) {
          const uint32_t i = 4 * lane + k;
          L val = (L)(sec[k] + lmid<L>());
          uint32_t par = 0xffffffffu;
          if (i < prim_cnt) {
            uint32_t lb = dlat[i];
            if (lb > window_n) lb = 1;   // (already flagged)
            if (lb == 0) { }
            else if (lb <= i) par = i - lb;
            else {
              const int64_t jsrc = (int64_t)(state_n + kbase + i) - (int64_t)lb;
              if (jsrc >= 0) val = (L)(val + __hip_atomic_load(&sh[jsrc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
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
        for (int k = 0; k < 4; k++) { const uint32_t i = 4 * lane + k; if (i < prim_cnt) sh[state_n + kbase + i] = scratch[i]; }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
    } else {
      // join + coalesced store: 4 contiguous numbers per lane
      if (raw_hist) {  // second pass: the primary latents are what the first pass left in dst (the secondary's, if delta'd too, in its scratch)
        for (int k = 0; k < 4; k++) { const uint32_t i = 4 * lane + k; prim[k] = i < batch_n ? __hip_atomic_load(&dst[j0 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (L)0; }
        if (dk[2] == kDeltaLookback) { for (int k = 0; k < 4; k++) { const uint32_t i = 4 * lane + k; sec[k] = i < batch_n ? __hip_atomic_load((L PCO_GLOBAL*)pp.sec_hist + j0 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (L)0; } }
      }
      L outv[4];
#pragma unroll
      for (int k = 0; k < 4; k++) outv[k] = join_one<L>(mode_kind, num_kind, mode_base, mode_k, prim[k], sec[k]);
      const uint32_t i0 = 4 * lane;
      L PCO_GLOBAL* o = dst + j0 + i0;
      if (i0 + 4 <= batch_n && (((uintptr_t)o) & 15) == 0) {
        if constexpr (sizeof(L) == 8) {
          typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
          u64x2 PCO_GLOBAL* p = (u64x2 PCO_GLOBAL*)o;
          u64x2 a; a.x = outv[0]; a.y = outv[1]; u64x2 b; b.x = outv[2]; b.y = outv[3];
          p[0] = a; p[1] = b;
        } else if constexpr (sizeof(L) == 4) {
          typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
          u32x4 a; a.x = outv[0]; a.y = outv[1]; a.z = outv[2]; a.w = outv[3];
          *(u32x4 PCO_GLOBAL*)o = a;
        } else { for (int k = 0; k < 4; k++) o[k] = outv[k]; }
      } else {
        for (int k = 0; k < 4; k++) if (i0 + k < batch_n) o[k] = outv[k];
      }
    }
    n_remaining -= batch_n;
  }
  }
  if (uni(wave_or_u32(lb_oob))) { status = PCO_GFX_CORRUPTION; return; }
  // trailing bits of the page must be zero (page_decompressor.rs:184-188: checked by the call that reads the page's last batch)
  mr.bit = bitpos;
  if (!mr.drain_empty_byte()) status = PCO_GFX_CORRUPTION;
  if (!mr.in_bounds()) status = PCO_GFX_INSUFFICIENT_DATA;
  if (status && pp.progress && n_pass == 1) *pp.progress = 1u + (n == 0 ? 0u : ((n - 1) / kBatchN) * kBatchN);
}

// Dict mode page (mode/dict.rs:70-90): the primary variable holds u32 indices into the chunk's dictionary (ChunkMeta, uncompressed, in
// src), under any delta encoding (None, Consecutive, Conv1 on the u32 indices; Lookback with its own u32 delta variable).
// num = from_latent_ordered(dict[index]); an index beyond the dictionary is corruption.  Lookback keeps its history -- the indices --
// in dst and maps them through the dictionary in place once the page is decoded (decode_chunk has made sure an index fits a number).
template <class L, bool kLds>
__device__ __noinline__ void decode_page_body_dict(gcptr_u8 src, uint64_t src_len, MetaReader& mr, tptr<kLds, uint8_t> tbl,
                                                  const PageParams& pp, L PCO_GLOBAL* dst, uint32_t& status) {
  const uint32_t lane = lane_id();
  uint8_t PCO_LDS* smem = lds_base();
  uint32_t PCO_LDS* moments = (uint32_t PCO_LDS*)(smem + kLdsMomOff);
  uint32_t PCO_LDS* dlat = (uint32_t PCO_LDS*)(smem + kLdsDlatOff);
  uint32_t PCO_LDS* scratch = (uint32_t PCO_LDS*)(smem + kLdsScratchOff);
  uint32_t PCO_LDS* parent = (uint32_t PCO_LDS*)(smem + kLdsParentOff);
  VarInfo PCO_LDS* vinfo = (VarInfo PCO_LDS*)(smem + kLdsVarOff);
  const uint32_t n = pp.n, num_kind = pp.num_kind;
  const uint32_t dk = uni(vinfo[1].delta_kind), dord = uni(vinfo[1].delta_order);
  const bool lookback = dk == kDeltaLookback;
  const uint32_t window_n = 1u << uni(vinfo[1].window_n_log);
  uint32_t n_bins[2], max_ob[2], asl[2], off_nodes[2], off_lower[2], off_ob[2];
#pragma unroll
  for (int vi = 0; vi < 2; vi++) {
    n_bins[vi] = uni(vinfo[vi].n_bins); max_ob[vi] = uni(vinfo[vi].max_ob); asl[vi] = uni(vinfo[vi].ans_size_log);
    off_nodes[vi] = uni(vinfo[vi].off_nodes); off_lower[vi] = uni(vinfo[vi].off_lower); off_ob[vi] = uni(vinfo[vi].off_ob);
  }
  const uint32_t nlps = (dk == kDeltaConsecutive || dk == kDeltaConv1) ? dord : (lookback ? (1u << uni(vinfo[1].state_n_log)) : 0u);
  const uint32_t state_n = lookback ? nlps : 0u;
  // ---- page meta (metadata/page.rs:36-57): [delta variable: 4 tANS states] then [primary: delta state, 4 tANS states] ----
  uint32_t st[2] = {0, 0};
#pragma unroll
  for (int vi = 0; vi < 2; vi++) {
    if (vi == 0 && !lookback) continue;
    if (vi == 1) for (uint32_t i = 0; i < nlps; i++) {
      const uint32_t x = (uint32_t)mr.read(32);
      if (dk == kDeltaConsecutive) { if (lane == 0 && i < 8) moments[i] = x; }
      else if (dk == kDeltaConv1) { if (lane == 0 && i < 32) scratch[i] = x; }
      else if (i < n && mr.in_bounds()) { if (lane == 0) dst[i] = (L)x; }   // lookback state = the first state_n indices
    }
    uint32_t mine = 0;
    for (uint32_t j = 0; j < 4; j++) { const uint32_t s = (uint32_t)mr.read(asl[vi]); if (lane == j) mine = s; }
    st[vi] = mine;
  }
  if (!mr.drain_empty_byte()) { if (mr.in_bounds()) status = PCO_GFX_CORRUPTION; }
  if (!mr.in_bounds()) { status = PCO_GFX_INSUFFICIENT_DATA; return; }
  if (status) return;
  if (n > nlps && (n_bins[1] == 0 || (lookback && n_bins[0] == 0))) { status = PCO_GFX_CORRUPTION; return; }
  wave_sync_lds();
  uint64_t bitpos = mr.bit;
  uint32_t n_remaining = n, oob = 0, lb_oob = 0;
  constexpr uint32_t kBytes = sizeof(L);
  auto dict_value = [&](uint32_t index) -> L {
    L v = 0;
    gcptr_u8 p = pp.dict_src + pp.dict_byte + (uint64_t)index * kBytes;
    for (uint32_t b = 0; b < kBytes; b++) v |= (L)((L)p[b] << (8 * b));   // (the dictionary sits at an arbitrary byte offset)
    return from_latent_ordered<L>(v, num_kind);
  };
  for (uint32_t j0 = 0; j0 < n; j0 += kBatchN) {
    const uint32_t batch_n = n_remaining < kBatchN ? n_remaining : kBatchN;
    const uint32_t rem = n_remaining > nlps ? n_remaining - nlps : 0, cnt = rem < kBatchN ? rem : kBatchN;
    uint32_t idx[4] = {0, 0, 0, 0};
#pragma unroll
    for (int vi = 0; vi < 2; vi++) {
      if (vi == 0 && !lookback) continue;
      const uint32_t c = vi == 0 ? (cnt < batch_n ? cnt : batch_n) : cnt;
      if (c == 0) { if (vi == 0) for (int k = 0; k < 4; k++) dlat[4 * lane + k] = 0u; continue; }
      uint32_t tmp[4];
      const bool single_bin = n_bins[vi] <= 1;
      if (!single_bin) bitpos = walk_ans<kLds>(src, src_len, bitpos, c, (tptr<kLds, uint32_t>)(tbl + off_nodes[vi]), st[vi]);
      tptr<kLds, uint32_t> lw = (tptr<kLds, uint32_t>)(tbl + off_lower[vi]);
      if (max_ob[vi] != 0 || !single_bin) bitpos = unpack_offsets<uint32_t, kLds>(src, src_len, bitpos, c, lw, tbl + off_ob[vi], single_bin, tmp);
      else { const uint32_t l0 = lw[0]; for (int k = 0; k < 4; k++) tmp[k] = 4 * lane + k < c ? l0 : 0u; }
      if (bitpos > src_len * 8) { status = PCO_GFX_INSUFFICIENT_DATA; return; }
      if (vi == 0) { for (int k = 0; k < 4; k++) dlat[4 * lane + k] = 4 * lane + k < c ? tmp[k] : 0u; }
      else { for (int k = 0; k < 4; k++) idx[k] = tmp[k]; }
    }
    if (dk == kDeltaConsecutive) consecutive_decode<uint32_t>(idx, dord, moments);
    else if (dk == kDeltaConv1) conv1_decode<uint32_t>(idx, batch_n, dord, pp.conv_quant);
    if (lookback) {
      // F[state_n + k] = delta_k + MID + F[state_n + k - lb_k] (delta/lookback.rs:200-246), the indices F living in dst
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      __builtin_amdgcn_wave_barrier();
      const uint64_t kbase = (uint64_t)j0;
      for (int k = 0; k < 4; k++) {
        const uint32_t i = 4 * lane + k;
        uint32_t val = idx[k] + 0x80000000u, par = 0xffffffffu;
        if (i < cnt) {
          uint32_t lb = dlat[i];
          if (lb > window_n) { lb_oob = 1; lb = 1; }
          if (lb == 0) {
            // (the slot's stale window content in the reference; only reachable from corrupt bins, which ChunkMeta validation rejects)
          } else if (lb <= i) par = i - lb;
          else {
            const int64_t jsrc = (int64_t)(state_n + kbase + i) - (int64_t)lb;
            if (jsrc >= 0) val += (uint32_t)__hip_atomic_load(&dst[jsrc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        scratch[i] = val; parent[i] = par;
      }
      wave_sync_lds();
      for (int round = 0; round < 8; round++) {  // pointer jumping: resolves in-batch chains of length <= 256
        uint32_t nv[4], np[4];
        for (int k = 0; k < 4; k++) {
          const uint32_t i = 4 * lane + k; const uint32_t p = parent[i];
          nv[k] = scratch[i]; np[k] = p;
          if (p != 0xffffffffu) { nv[k] += scratch[p]; np[k] = parent[p]; }
        }
        wave_sync_lds();
        for (int k = 0; k < 4; k++) { const uint32_t i = 4 * lane + k; scratch[i] = nv[k]; parent[i] = np[k]; }
        wave_sync_lds();
      }
      for (int k = 0; k < 4; k++) {
        const uint32_t i = 4 * lane + k;
        if (i < cnt) { const uint32_t f = scratch[i]; if (f >= pp.dict_n) oob = 1; dst[state_n + kbase + i] = (L)f; }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t i = 4 * lane + k;
        if (i < batch_n) { if (idx[k] >= pp.dict_n) oob = 1; else dst[j0 + i] = dict_value(idx[k]); }
      }
    }
    n_remaining -= batch_n;
  }
  if (uni(wave_or_u32(lb_oob))) { status = PCO_GFX_CORRUPTION; return; }
  if (lookback) {   // the page's indices are complete: map them through the dictionary in place
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (uint32_t j = lane; j < n; j += 64) {
      const L f = __hip_atomic_load(&dst[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((uint64_t)f >= (uint64_t)pp.dict_n) oob = 1; else dst[j] = dict_value((uint32_t)f);
    }
  }
  if (uni(wave_or_u32(oob))) { status = PCO_GFX_CORRUPTION; return; }
  mr.bit = bitpos;
  if (!mr.drain_empty_byte()) status = PCO_GFX_CORRUPTION;
  if (!mr.in_bounds()) status = PCO_GFX_INSUFFICIENT_DATA;
}

template <class L>
__device__ __noinline__ void decode_chunk(gcptr_u8 src, uint64_t src_len, MetaReader& mr, uint32_t lds_table_budget,
                                         gptr_u8 tbl_ws, uint32_t format_major, uint32_t dtype, uint32_t n, L PCO_GLOBAL* dst, uint32_t& status,
                                         bool meta_only, void PCO_GLOBAL* sec_hist = nullptr, uint32_t need_hist_status = PCO_GFX_UNSUPPORTED, uint32_t* progress = nullptr,
                                         gcptr_u8 page_src = nullptr, uint64_t page_len = 0) {   // page_src: the page lives in a buffer of its own (src holds the ChunkMeta only); mr is re-seated on it
  const uint32_t lane = lane_id();
  const uint32_t num_kind = dtype_kind(dtype);
  constexpr uint32_t LB = LBits<L>::v;
  VarInfo PCO_LDS* vinfo = (VarInfo PCO_LDS*)(lds_base() + kLdsVarOff);
  // ---- ChunkMeta (metadata/chunk.rs:127-174) ----
  const uint32_t mode_kind = (uint32_t)mr.read(kBitsModeVariant);
  L mode_base = 0; uint32_t mode_k = 0, dict_n = 0, conv_quant = 0; uint64_t dict_byte = 0;
  if (mode_kind == kIntMult) {
    if (format_major == 0) { status = PCO_GFX_CORRUPTION; return; }
    mode_base = (L)mr.read(LB);
  } else if (mode_kind == kFloatMult) mode_base = (L)mr.read(LB);
  else if (mode_kind == kFloatQuant) mode_k = (uint32_t)mr.read(kBitsQuantK);
  else if (mode_kind == kDict) {   // metadata/mode.rs:138-165: 25-bit length, byte alignment, then the dictionary's latents, uncompressed
    dict_n = (uint32_t)mr.read(kBitsDictLen);
    if (!mr.in_bounds()) { status = PCO_GFX_INSUFFICIENT_DATA; return; }
    if (!mr.drain_empty_byte()) { status = PCO_GFX_CORRUPTION; return; }
    dict_byte = mr.bit >> 3;
    mr.bit += (uint64_t)dict_n * LB;
  }
  else if (mode_kind != kClassic) { status = mr.in_bounds() ? PCO_GFX_CORRUPTION : PCO_GFX_INSUFFICIENT_DATA; return; }
  if (!mr.in_bounds()) { status = PCO_GFX_INSUFFICIENT_DATA; return; }
  const bool dict = mode_kind == kDict;
  // delta encoding (metadata/delta_encoding.rs:118-202)
  uint32_t dkind = kDeltaNone, dorder = 0, wlog = 0, slog = 0; bool sec_uses_delta = false;
  if (format_major < 3) {
    dorder = (uint32_t)mr.read(kBitsDeltaOrder);
    if (dorder) dkind = kDeltaConsecutive;
  } else {
    const uint32_t variant = (uint32_t)mr.read(kBitsDeltaVariant);
    if (variant == 1) {
      dorder = (uint32_t)mr.read(kBitsDeltaOrder);
      if (dorder == 0) { status = PCO_GFX_CORRUPTION; return; }
      dkind = kDeltaConsecutive; sec_uses_delta = mr.read(1) != 0;
    } else if (variant == 2) {
      wlog = 1 + (uint32_t)mr.read(kBitsLookbackWindowLog); slog = (uint32_t)mr.read(kBitsLookbackStateLog);
      if (wlog > kMaxLookbackWindowLog || slog > wlog) { status = PCO_GFX_CORRUPTION; return; }
      dkind = kDeltaLookback; sec_uses_delta = mr.read(1) != 0;
    } else if (variant == 3) {   // Conv1 (metadata/delta_encoding.rs): 5-bit quantization, 64-bit bias, 5-bit weight count - 1, 32-bit weights
      conv_quant = (uint32_t)mr.read(5);
      const uint64_t bias = mr.read(64) ^ ((uint64_t)1 << 63);
      dorder = 1 + (uint32_t)mr.read(5);
      int64_t PCO_LDS* cw = (int64_t PCO_LDS*)(lds_base() + kLdsConvOff);
      if (lane == 0) cw[0] = (int64_t)bias;
      for (uint32_t k = 0; k < dorder; k++) { const uint32_t w = (uint32_t)mr.read(32) ^ 0x80000000u; if (lane == 0) cw[1 + k] = (int64_t)(int32_t)w; }
      dkind = kDeltaConv1;
    }
    else if (variant != 0) { status = PCO_GFX_CORRUPTION; return; }
  }
  if (!mr.in_bounds()) { status = PCO_GFX_INSUFFICIENT_DATA; return; }

  const uint32_t present[3] = {dkind == kDeltaLookback ? 1u : 0u, 1u,
                               (mode_kind == kIntMult || mode_kind == kFloatMult || mode_kind == kFloatQuant) ? 1u : 0u};
  // First pass over the per-variable headers only: decides where the tANS tables live (LDS or global).
  uint32_t total_tbl = 0;
  {
    MetaReader peek = mr;
#pragma unroll
    for (int vi = 0; vi < 3; vi++) {
      VarInfo v{};
      v.present = present[vi]; v.latent_bits = (vi == 0 || (vi == 1 && dict)) ? 32u : LB;   // Dict: the primary holds u32 dictionary indices (mode.rs:197-202)
      if (vi == 1 || (vi == 2 && sec_uses_delta)) { v.delta_kind = dkind; v.delta_order = dorder; v.window_n_log = wlog; v.state_n_log = slog; }
      if (present[vi]) {
        const uint32_t a = (uint32_t)peek.read(kBitsAnsSizeLog), nb = (uint32_t)peek.read(kBitsNBins);
        if (!peek.in_bounds()) { status = PCO_GFX_INSUFFICIENT_DATA; return; }
        if ((1u << a) < nb || (nb == 1 && a > 0) || a > kMaxAnsBits) { status = PCO_GFX_CORRUPTION; return; }
        v.ans_size_log = a; v.n_bins = nb;
        const uint32_t lbytes = v.latent_bits / 8;
        v.off_nodes = total_tbl;
        v.off_lower = total_tbl + (4u << a);
        v.off_ob = v.off_lower + ((nb * lbytes + 7u) & ~7u);
        v.off_cum = v.off_ob + ((nb + 7u) & ~7u);
        total_tbl += var_table_bytes(a, nb, lbytes);
        peek.bit += (uint64_t)nb * (a + v.latent_bits + offset_bits_bits(v.latent_bits));
      }
      if (lane == 0) {
        uint32_t PCO_LDS* p = (uint32_t PCO_LDS*)(vinfo + vi);
        const uint32_t* q = (const uint32_t*)&v;
        for (int w = 0; w < 16; w++) p[w] = q[w];
      }
    }
  }
  wave_sync_lds();
  const bool lds_tables = total_tbl <= lds_table_budget;
  // (no global table scratch in this pass: a synchronous call hands the task back and runs it again with the scratch, like a task that
  //  needs a second history buffer -- tANS tables beyond the LDS budget are rare, 832 KB of scratch per block for every call were not)
  if (!lds_tables && tbl_ws == nullptr) { status = need_hist_status == (uint32_t)PCO_GFX_UNSUPPORTED ? (uint32_t)PCO_GFX_DEVICE_ERROR : need_hist_status; return; }
  uint8_t PCO_LDS* tbl_lds = lds_base() + kLdsFixed;
#pragma unroll
  for (int vi = 0; vi < 3; vi++) {
    if (!present[vi]) continue;
    mr.bit += kBitsAnsSizeLog + kBitsNBins;
    bool ok;
    const bool u32_var = vi == 0 || (vi == 1 && dict);
    if (lds_tables) ok = u32_var ? build_var_tables<uint32_t, true>(tbl_lds, vi, mr, status) : build_var_tables<L, true>(tbl_lds, vi, mr, status);
    else ok = u32_var ? build_var_tables<uint32_t, false>(tbl_ws, vi, mr, status) : build_var_tables<L, false>(tbl_ws, vi, mr, status);
    if (!ok) return;
  }
  if (!mr.drain_empty_byte()) { if (mr.in_bounds()) { status = PCO_GFX_CORRUPTION; return; } }
  if (!mr.in_bounds()) { status = PCO_GFX_INSUFFICIENT_DATA; return; }
  // ChunkMeta::new validation (metadata/chunk.rs:38-57): lookback bins must lie in [1, window_n]
  if (dkind == kDeltaLookback) {
    const uint32_t nb0 = uni(vinfo[0].n_bins), off0 = uni(vinfo[0].off_lower);
    uint32_t bad = 0;
    for (uint32_t b = lane; b < nb0; b += 64) {
      const uint32_t lw = lds_tables ? ((uint32_t PCO_LDS*)(tbl_lds + off0))[b] : ((uint32_t PCO_GLOBAL*)(tbl_ws + off0))[b];
      if (lw < 1 || lw > (1u << wlog)) bad = 1;
    }
    if (uni(wave_or_u32(bad))) { status = PCO_GFX_CORRUPTION; return; }
    // a delta'd secondary variable has a history of its own, which needs n latents of scratch: a task that came without is handed back
    // (synchronous calls come again with the scratch; no encoder writes this combination, wrapped/chunk_compressor.rs:343,384)
    if (sec_uses_delta && present[2] && !meta_only && sec_hist == nullptr) { status = need_hist_status; return; }
  }
  // mode validity for the number type (data_types/unsigned.rs:65-71, float.rs:377-390)
  {
    bool valid = true;
    if (mode_kind == kIntMult) valid = num_kind != kFloat && mode_base > 0;
    else if (mode_kind == kFloatQuant) { const uint32_t prec = LB == 64 ? 52 : (LB == 32 ? 23 : 10); valid = num_kind == kFloat && mode_k > 0 && mode_k <= prec; }
    else if (mode_kind == kFloatMult) {
      if (num_kind != kFloat) valid = false;
      else if constexpr (sizeof(L) >= 4) {
        typedef typename FloatOf<L>::F F;
        const F b = bits_to_float(from_latent_ordered<L>(mode_base, kFloat));
        valid = isfinite(b) && b != (F)0;
      } else if constexpr (sizeof(L) == 2) { const uint32_t hb = (uint32_t)from_latent_ordered<L>(mode_base, kFloat); valid = (hb & 0x7c00u) != 0x7c00u && (hb & 0x7fffu) != 0; }
      else valid = false;
    }
    if (!valid) { status = PCO_GFX_CORRUPTION; return; }
  }
  if (dkind == kDeltaConv1) {   // ChunkMeta::new validation (metadata/chunk.rs:58-94); the primary variable's latent type decides (u32 in Dict mode)
    const uint32_t plb = dict ? 32u : LB;
    if (plb > 32) { status = PCO_GFX_CORRUPTION; return; }   // "Conv1 delta encodings are not supported on types larger than 32 bits"
    const uint32_t conv_bits = plb == 32 ? 64u : 2 * plb;   // Conv = i16 / i32 / i64 (data_types/unsigned.rs:132-134)
    const uint32_t max_quant = conv_bits - 1 < 31u ? conv_bits - 1 : 31u;
    if (conv_quant > max_quant) { status = PCO_GFX_CORRUPTION; return; }
    const int64_t PCO_LDS* cw = (const int64_t PCO_LDS*)(lds_base() + kLdsConvOff);
    double sum_abs = 0.0;
    for (uint32_t k = 0; k < dorder; k++) { const int64_t wk = cw[1 + k]; sum_abs += (double)(wk < 0 ? -wk : wk); }
    const double max_pred = fabs((double)cw[0]) + ldexp(1.0, (int)plb) * sum_abs;
    if (max_pred >= ldexp(1.0, (int)conv_bits - 1)) { status = PCO_GFX_CORRUPTION; return; }   // "weights and bias risk overflowing"
  }
  // Dict + Lookback keeps the u32 indices in dst until the page is done: they must fit a number.  A dictionary of distinct latents
  // (the only kind an encoder writes, mode/dict.rs:12-33) always does; a padded one on an 8- / 16-bit type is refused.
  if (dict && dkind == kDeltaLookback && LB < 32 && dict_n > (1u << LB)) { status = PCO_GFX_UNSUPPORTED; return; }
  if (meta_only) return;
  PageParams pp{mode_kind, mode_k, num_kind, n, (uint64_t)mode_base, dict_byte, dict_n, dorder, conv_quant, sec_hist, src, progress};
  gcptr_u8 body_src = src; uint64_t body_len = src_len;
  if (page_src != nullptr) { body_src = page_src; body_len = page_len; mr = MetaReader{page_src, page_len, 0}; }   // wrapped/page_decompressor.rs:82-113: a page is read from its own source
  if (dict) {
    if (lds_tables) decode_page_body_dict<L, true>(body_src, body_len, mr, tbl_lds, pp, dst, status);
    else decode_page_body_dict<L, false>(body_src, body_len, mr, tbl_ws, pp, dst, status);
    return;
  }
  if (lds_tables) decode_page_body<L, true>(body_src, body_len, mr, tbl_lds, pp, dst, status);
  else decode_page_body<L, false>(body_src, body_len, mr, tbl_ws, pp, dst, status);
}

// One wave per task; a task is a stream of >= 1 standalone chunks of number width sizeof(L).
#ifndef PCO_DEC_MIN_WAVES
#define PCO_DEC_MIN_WAVES 4   // waves per SIMD the register allocator must leave room for
#endif
template <class L>
__global__ __launch_bounds__(64, PCO_DEC_MIN_WAVES) void pco_decode_kernel(const PcoGfxDecodeTask* tasks, PcoGfxTaskResult* results, const uint32_t* task_ids,
                                                        uint32_t n_ids, uint32_t lds_table_budget, uint8_t* tbl_ws_base,
                                                        const uint32_t* only_if_status, uint32_t status_stride_u32, uint32_t status_value,
                                                        uint8_t* hist_base, const uint64_t* hist_off, uint32_t need_hist_status, const MetaRef* metas) {
  const uint32_t lane = lane_id();
  for (uint32_t bi = blockIdx.x; bi < n_ids; bi += gridDim.x) {
    const uint32_t ti = task_ids ? task_ids[bi] : bi;
    // when chained after the two-kernel fast path: only finish the tasks it handed over
    if (only_if_status && uni(only_if_status[(uint64_t)ti * status_stride_u32]) != status_value) continue;
    const PcoGfxDecodeTask task = tasks[ti];
    gcptr_u8 src = (gcptr_u8)task.src;
    const uint64_t src_len = uni((uint64_t)task.src_len);
    const uint32_t dtype = uni(task.dtype), flags = uni(task.flags);
    const uint64_t dst_cap = uni((uint64_t)task.dst_cap);
    MetaReader mr{src, src_len, 0};
    uint32_t status = PCO_GFX_OK;
    uint64_t n_out = 0;
    uint32_t format_major = 4, uniform_type = 0;
    if (dtype_bits(dtype) != (int)LBits<L>::v) status = PCO_GFX_INVALID_ARGUMENT;
    if (!status && (flags & (PCO_GFX_TASK_WRAPPED_PAGE | PCO_GFX_TASK_META_ONLY))) {
      // wrapped surface: src = ChunkMeta [+ one page of exactly dst_cap numbers]; the format version comes from the caller
      format_major = (flags >> 8) & 0xffu;
      const bool meta_only = (flags & PCO_GFX_TASK_META_ONLY) != 0;
      const uint32_t n = meta_only ? 1u : (uint32_t)dst_cap;
      uint32_t aux_body = 0;
      if (!meta_only && (dst_cap == 0 || dst_cap > kMaxEntries)) status = PCO_GFX_INVALID_ARGUMENT;
      if (!status) {
        gptr_u8 tbl_ws = tbl_ws_base ? (gptr_u8)tbl_ws_base + (uint64_t)blockIdx.x * kTblWsBytes : (gptr_u8) nullptr;
        uint32_t progress = 0;
        // PcoGfxPageTask (pco_gfx_decompress_pages): the ChunkMeta in a buffer of its own, shared by the chunk's pages; src is the page alone
        gcptr_u8 meta_p = metas != nullptr && !meta_only ? (gcptr_u8)metas[ti].p : (gcptr_u8) nullptr;
        const uint64_t meta_len = meta_p != nullptr ? uni((uint64_t)metas[ti].len) : 0;
        if (meta_p != nullptr) mr = MetaReader{meta_p, meta_len, 0};
        decode_chunk<L>(meta_p != nullptr ? meta_p : src, meta_p != nullptr ? meta_len : src_len, mr, lds_table_budget, tbl_ws, format_major, dtype, n, (L PCO_GLOBAL*)task.dst, status, meta_only,
                        hist_base ? (void PCO_GLOBAL*)(hist_base + hist_off[bi]) : (void PCO_GLOBAL*)nullptr, need_hist_status, &progress,
                        meta_p != nullptr ? src : (gcptr_u8) nullptr, src_len);
        status = uni(status); progress = uni(progress);
        if (!status && !meta_only) n_out = n;
        // a page that failed inside its body: the batches before the failing one are in dst (page_decompressor.rs:115-221 hands them out
        // and fails on the call that reaches the bad batch) -- n_out says how many numbers that is, aux bit 0 that the body was reached
        if (status && progress) { n_out = progress - 1; aux_body = 1; }
      }
      if (lane == 0) { PcoGfxTaskResult r; r.n_out = n_out; r.consumed = mr.bit >> 3; r.status = status; r.aux = aux_body; results[ti] = r; }
      wave_sync_lds();
      continue;
    }
    if (!status && (flags & PCO_GFX_TASK_HAS_FILE_HEADER)) {
      // standalone/decompressor.rs:85-137
      const uint32_t magic = (uint32_t)mr.read(32);
      if (!mr.in_bounds()) status = PCO_GFX_INSUFFICIENT_DATA;
      else if (magic != 0x216f6370u) status = PCO_GFX_CORRUPTION;
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
          if (format_major > 4) status = PCO_GFX_CORRUPTION;
          else if (!mr.in_bounds()) status = PCO_GFX_INSUFFICIENT_DATA;
        }
      }
    }
    if (!status && !(flags & PCO_GFX_TASK_HAS_FILE_HEADER) && (flags & PCO_GFX_TASK_ONE_CHUNK) && ((flags >> 8) & 0xffu) != 0) format_major = (flags >> 8) & 0xffu;
    uint32_t more = 0;
    while (!status) {
      // chunk preamble (standalone/decompressor.rs:190-231)
      if (!(flags & PCO_GFX_TASK_HAS_FILE_HEADER) && (mr.bit >> 3) >= src_len) break;
      const uint32_t tb = (uint32_t)mr.read(8);
      if (!mr.in_bounds()) { status = PCO_GFX_INSUFFICIENT_DATA; break; }
      if (tb == 0) break;  // terminator
      if ((uniform_type && uniform_type != tb) || tb != dtype) { status = PCO_GFX_CORRUPTION; break; }
      const uint32_t n = (uint32_t)mr.read(kBitsNEntries) + 1;
      if (!mr.in_bounds()) { status = PCO_GFX_INSUFFICIENT_DATA; break; }
      if (n_out + n > dst_cap) { status = PCO_GFX_INVALID_ARGUMENT; break; }
      gptr_u8 tbl_ws = tbl_ws_base ? (gptr_u8)tbl_ws_base + (uint64_t)blockIdx.x * kTblWsBytes : (gptr_u8) nullptr;
      decode_chunk<L>(src, src_len, mr, lds_table_budget, tbl_ws, format_major, dtype, n, (L PCO_GLOBAL*)task.dst + n_out, status, false,
                      hist_base ? (void PCO_GLOBAL*)((L PCO_GLOBAL*)(hist_base + hist_off[bi]) + n_out) : (void PCO_GLOBAL*)nullptr, need_hist_status);
      status = uni(status);
      if (!status) n_out += n;
      wave_sync_lds();
      if (!status && (flags & PCO_GFX_TASK_ONE_CHUNK)) {   // this chunk only: say whether another follows, take the terminator if not
        const uint64_t byte = mr.bit >> 3;
        if (byte < src_len && uni((uint32_t)src[byte]) != 0) more = 1;
        else if (byte < src_len) { mr.bit += 8; more = 2; }   // aux bit 1: the terminator was there and is consumed
        else if (flags & PCO_GFX_TASK_HAS_FILE_HEADER) status = PCO_GFX_INSUFFICIENT_DATA;
        break;
      }
    }
    if (lane == 0) {
      PcoGfxTaskResult r; r.n_out = n_out; r.consumed = mr.bit >> 3; r.status = status; r.aux = more;
      results[ti] = r;
    }
  }
}

}  // namespace pcogfx
