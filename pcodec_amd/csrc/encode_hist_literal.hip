// encode_hist_literal.hip -- the STRICT histogram: histograms.rs:60-298 + sort_utils.rs replayed literally, pivot by pivot.
//
// The reference's equal-count histogram is a quickselect recursion whose RESULT is a function of the sorted multiset of the latents
// -- except on one branch: after 1 + log2(n + 1) bad pivots on one recursion path it heapsorts the node and applies `apply_sorted`
// (histograms.rs:248-258), whose treatment of runs of equal values at a bin end differs from the quickselect path's.  Whether that
// branch runs depends on the ORDER of the latents: the pivots are read at fixed positions of an array that every earlier partition
// (sort_utils.rs:109-126, a Lomuto pass) and every `break_patterns` (three swaps from a xorshift seeded by the length) has permuted.
// The fast histogram kernels (enc_hist_kernel, enc_hist_wide_kernel, enc_hist_select_kernel, ...) compute the multiset function and
// are therefore bit-identical to the reference wherever that branch does not run (0 of 68 200 chunks in the census; it takes an
// order built against the pivot rule).  This kernel closes the gap for callers that ask for it (PCO_GFX_CFG_STRICT_HISTOGRAM): one
// wave per chunk replays the recursion on a private copy of the variable's stored latents and overwrites the bins the fast kernels
// left in the plan region, so that the bytes are the reference's on EVERY order.
//
// What is parallel in it: the Lomuto pass has a closed form per 64-element tile.  With L(q) = number of elements < pivot among the
// first q, the k-th smaller element lands at position k (stable), and position q of the not-smaller block receives, at step q,
//     C(q) = v[q]        if L(q) == q            (nothing not-smaller seen yet)
//          = v[q - 1]    if v[q - 1] >= pivot    (step q - 1 put v[q - 1] at the block's front, step q moves it to the block's end)
//          = C(L(q))     otherwise               (the front is the element step L(q) put at position L(q))
// -- a prefix count (one ballot), one gather of elements the wave itself stored at least two tiles ago, and, only while the block
// is shorter than a tile, pointer jumping inside the tile; the position the block's front occupies when the pass ends gets the
// last not-smaller element.  (Checked against the literal loop on random arrays by scripts/lomuto_tile_model.py.)  Pivot choice,
// `break_patterns`, the bound bookkeeping and the builder (apply_incomplete / complete_bin / apply_constant_run) are the reference's
// scalar logic on wave-uniform values; a leaf with loose bounds takes its minimum / maximum by a wave reduction.  The heapsort of a
// node is replaced by an LSD radix sort (its result, the sorted node, is the same) and `apply_sorted`'s run scans by binary searches.
//
// Cost: ~ (1 + log2(bins)) passes over a private copy of the variable: at 8192 chunks the launch is bound by HBM at several times
// the whole fast encode -- which is why strict mode is opt-in.
#pragma once

namespace pcogfx {

constexpr uint32_t kLitStackCap = 224;                                // pending right siblings: <= 5.2 log2(n) + limit + 16 for n <= 2^24
constexpr uint32_t kLitLdsStack = 0;                                  // {u32 lo | flags, u32 len, u64 lb, u64 ub}[kLitStackCap]
constexpr uint32_t kLitLdsCnt = kLitStackCap * 24;                    // u32[256] radix counters
constexpr uint32_t kLitLdsBytes = kLitLdsCnt + 256 * 4;

__device__ __forceinline__ void lit_sync() {   // one wave per block: program order between the lanes' global accesses
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  __builtin_amdgcn_wave_barrier();
}
template <class L> __device__ __forceinline__ L lit_uni(L v) {
  if constexpr (sizeof(L) == 8) return (L)uni((uint64_t)v);
  else return (L)uni((uint32_t)v);
}
template <class L> __device__ __forceinline__ L lit_readlane(L v, uint32_t src) {   // src wave-uniform
  if constexpr (sizeof(L) == 8) return (L)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), (int)src) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)src));
  else return (L)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)src);
}

// the builder (histograms.rs:60-161): wave-uniform state, bins go straight to the plan region
template <class L> struct LitBuilder {
  uint64_t n; uint32_t bins_log;
  uint32_t n_applied = 0, next_avail = 0, n_hist = 0, inc_count = 0; bool has_inc = false; L inc_lower = 0, inc_upper = 0;
  PlanRef plan;
  __device__ __forceinline__ uint32_t bin_idx(uint32_t c) const { return (uint32_t)((((uint64_t)c) << bins_log) / n); }
  __device__ __forceinline__ uint32_t c_count(uint32_t b) const { return (uint32_t)((((uint64_t)b + 1) * n + ((uint64_t)1 << bins_log) - 1) >> bins_log); }
  __device__ __forceinline__ bool complete_bin(uint32_t b) {
    if (!has_inc) return false;
    next_avail = b + 1;
    if (n_hist < plan.cap && lane_id() == 0) { plan.hcount()[n_hist] = inc_count; plan.hlower()[n_hist] = (uint64_t)inc_lower; plan.hupper()[n_hist] = (uint64_t)inc_upper; }
    n_hist++; has_inc = false;
    return true;
  }
};

// minimum and / or maximum of a slice (histograms.rs:16-54), by the wave
template <class L> __device__ __forceinline__ void lit_min_max(const L PCO_GLOBAL* v, uint32_t len, L& mn, L& mx) {
  typedef typename std::conditional<sizeof(L) == 8, uint64_t, uint32_t>::type W;
  W a = (W)(L)~(L)0, b = 0;
  for (uint32_t i = lane_id(); i < len; i += 64) { const W x = (W)v[i]; a = x < a ? x : a; b = x > b ? x : b; }
  a = wave_butterfly(a, [](W x, W y) { return x < y ? x : y; });
  b = wave_butterfly(b, [](W x, W y) { return x > y ? x : y; });
  mn = (L)lit_uni(a); mx = (L)lit_uni(b);
}
// (K: the type of the keys in the scratch copy -- the latents themselves, or 16-bit latents relative to `ref`; bounds and bins are latents)
template <class K, class L>
__device__ __forceinline__ void lit_apply_incomplete(LitBuilder<L>& hb, const K PCO_GLOBAL* v, L ref, uint32_t len, bool lower_tight, L lower, bool upper_tight, L upper) {   // :82-106
  if (len == 0) return;
  L mn = lower, mx = upper;
  const bool need_min = !hb.has_inc && !lower_tight, need_max = !upper_tight;
  if (need_min || need_max) { K a, b; lit_min_max<K>(v, len, a, b); if (need_min) mn = (L)(ref + (L)a); if (need_max) mx = (L)(ref + (L)b); }
  if (hb.has_inc) { hb.inc_upper = mx; hb.inc_count += len; }
  else { hb.inc_lower = mn; hb.inc_upper = mx; hb.inc_count = len; hb.has_inc = true; }
  hb.n_applied += len;
}
template <class L>
__device__ __forceinline__ void lit_apply_constant_run(LitBuilder<L>& hb, uint32_t len, L value) {   // :142-161
  const uint32_t start = hb.n_applied, mid = start + len / 2, end = start + len;
  uint32_t b = hb.bin_idx(mid);
  if (b > hb.next_avail) { const uint32_t spare = b - 1; if (!hb.complete_bin(spare)) b = spare; }
  lit_apply_incomplete<L, L>(hb, nullptr, (L)0, len, true, value, true, value);
  if (end >= hb.c_count(b)) hb.complete_bin(b);
}

template <class L> __device__ __forceinline__ L lit_median3(L x, L y, L z) {   // sort3's middle (sort_utils.rs:31-35), by value
  if (y < x) { const L t = x; x = y; y = t; }
  if (z < y) { const L t = y; y = z; z = t; }
  if (y < x) { const L t = x; x = y; y = t; }
  return y;
}
template <class L> __device__ __forceinline__ L lit_choose_pivot(const L PCO_GLOBAL* v, uint32_t len) {   // sort_utils.rs:5-56
  const uint32_t a = len / 4, b = len / 2, c = (uint32_t)(((uint64_t)len * 3) / 4);
  if (len < 8) return lit_uni<L>(v[b]);
  // nine positions, one lane each (three below 50 elements): median of the three neighbourhood medians
  const uint32_t lane = lane_id(), grp = lane / 3, k = lane % 3;
  const uint32_t centre = grp == 0 ? a : (grp == 1 ? b : c);
  const bool wide = len >= 50;
  const uint32_t idx = wide ? centre + k - 1 : centre;
  const L x = lane < 9 ? v[idx] : (L)0;
  L m[3];
#pragma unroll
  for (uint32_t g = 0; g < 3; g++) {
    const L x0 = lit_readlane<L>(x, 3 * g), x1 = lit_readlane<L>(x, 3 * g + 1), x2 = lit_readlane<L>(x, 3 * g + 2);
    m[g] = wide ? lit_median3<L>(x0, x1, x2) : x1;
  }
  return lit_median3<L>(m[0], m[1], m[2]);
}
template <class L> __device__ __forceinline__ void lit_break_patterns(L PCO_GLOBAL* v, uint32_t len) {   // sort_utils.rs:61-105 (usize = u64)
  if (len < 8) return;
  uint64_t seed = len;
  uint32_t modulus = 1; while (modulus < len) modulus <<= 1;
  const uint32_t pos = len / 4 * 2;
  for (uint32_t i = 0; i < 3; i++) {
    uint64_t r = seed; r ^= r << 13; r ^= r >> 7; r ^= r << 17; seed = r;
    uint32_t other = (uint32_t)(r & (uint64_t)(modulus - 1));
    if (other >= len) other -= len;
    const uint32_t p = pos - 1 + i;
    const L x = lit_uni<L>(v[p]), y = lit_uni<L>(v[other]);
    lit_sync();
    if (lane_id() == 0) { v[p] = y; v[other] = x; }
    lit_sync();
  }
}

// sort_utils.rs:109-126, 64 steps at a time (see the header).  Returns the count on the left side.
template <class L> __device__ __forceinline__ uint32_t lit_partition(L PCO_GLOBAL* v, uint32_t len, L pivot) {
  const uint32_t lane = lane_id();
  const uint64_t below = ((uint64_t)1 << lane) - 1;
  uint32_t left0 = 0; L prev_val = 0; bool prev_lt = true;
  for (uint32_t P = 0; P < len; P += 64) {
    const uint32_t m = len - P < 64u ? len - P : 64u;
    const bool act = lane < m;
    const uint32_t q = P + lane;
    const L val = act ? v[q] : (L)0;
    const bool lt = act && val < pivot;
    const uint64_t bal = __ballot(lt);
    const uint32_t k = (uint32_t)__popcll(bal), rank = (uint32_t)__popcll(bal & below);
    const uint32_t Lq = left0 + rank, left1 = left0 + k;
    L pv = shfl_up<L>(val, 1); bool pl = lane > 0 && ((bal >> ((lane - 1) & 63u)) & 1u) != 0;
    if (lane == 0) { pv = prev_val; pl = prev_lt; }
    L C = val; uint32_t ptr = lane; bool from_mem = false;
    if (act && Lq != q) {
      if (!pl) C = pv;
      else if (Lq >= P) ptr = Lq - P;
      else from_mem = true;
    }
    if (from_mem) C = v[Lq];
    if (__ballot(ptr != lane) != 0) {   // the block of not-smaller elements starts inside this tile: C(q) = C(L(q)) by pointer jumping
#pragma unroll
      for (int r = 0; r < 6; r++) { const L c2 = shfl_idx<L>(C, (int)ptr); const uint32_t p2 = (uint32_t)__shfl((int)ptr, (int)ptr, 64); C = c2; ptr = p2; }
    }
    if (lt) v[Lq] = val;
    if (act && q >= left1) v[q] = C;
    prev_val = lit_readlane<L>(val, m - 1); prev_lt = ((bal >> (m - 1)) & 1u) != 0;
    left0 = left1;
    if (P + m - left1 < 128u) lit_sync();   // while the block is short the next tile's gather reads what this tile stored
  }
  if (len > 0 && !prev_lt && left0 < len && lane == 0) v[left0] = prev_val;
  lit_sync();
  return left0;
}

// the node [v, v + len) in ascending order (what the two heapsorts of histograms.rs:250-251 leave: everything left of the split is
// smaller than everything right of it).  LSD radix sort by one wave, 8-bit digits, ping-pong with `tmp`.
template <class L> __device__ void lit_sort(L PCO_GLOBAL* v, L PCO_GLOBAL* tmp, uint32_t len, uint32_t PCO_LDS* cnt) {
  const uint32_t lane = lane_id();
  const uint64_t below = ((uint64_t)1 << lane) - 1;
  L PCO_GLOBAL* src = v; L PCO_GLOBAL* dst = tmp;
  for (uint32_t pass = 0; pass < sizeof(L); pass++) {
    const uint32_t sh = 8 * pass;
    for (uint32_t i = lane; i < 256; i += 64) cnt[i] = 0;
    lit_sync();
    for (uint32_t i = lane; i < len; i += 64) atomicAdd((uint32_t*)&cnt[(uint32_t)((uint64_t)src[i] >> sh) & 255u], 1u);
    lit_sync();
    {  // exclusive prefix over the 256 counters: four per lane
      const uint32_t c0 = cnt[4 * lane], c1 = cnt[4 * lane + 1], c2 = cnt[4 * lane + 2], c3 = cnt[4 * lane + 3];
      const uint32_t s = c0 + c1 + c2 + c3, base = wave_incl_scan(s) - s;
      lit_sync();
      cnt[4 * lane] = base; cnt[4 * lane + 1] = base + c0; cnt[4 * lane + 2] = base + c0 + c1; cnt[4 * lane + 3] = base + c0 + c1 + c2;
    }
    lit_sync();
    for (uint32_t P = 0; P < len; P += 64) {
      const bool act = P + lane < len;
      const L x = act ? src[P + lane] : (L)0;
      const uint32_t d = (uint32_t)((uint64_t)x >> sh) & 255u;
      uint64_t same = __ballot(act);
#pragma unroll
      for (uint32_t b = 0; b < 8; b++) { const uint64_t bb = __ballot(((d >> b) & 1u) != 0); same &= ((d >> b) & 1u) ? bb : ~bb; }
      const uint32_t off = act ? cnt[d] : 0u;
      lit_sync();
      if (act) {
        dst[off + (uint32_t)__popcll(same & below)] = x;
        if ((same >> lane) >> 1 == 0) cnt[d] = off + (uint32_t)__popcll(same);   // the group's highest lane moves the cursor on
      }
      lit_sync();
    }
    L PCO_GLOBAL* t = src; src = dst; dst = t;
  }
  if (src != v) { for (uint32_t i = lane; i < len; i += 64) v[i] = src[i]; lit_sync(); }
}

// histograms.rs:164-206 on a sorted node; the scans over equal values are binary searches
template <class K, class L> __device__ void lit_apply_sorted(LitBuilder<L>& hb, const K PCO_GLOBAL* v, L ref, uint32_t len) {
  auto at = [&](uint32_t i) -> L { return (L)(ref + (L)lit_uni<K>(v[i])); };
  while (len > 0) {
    const uint32_t target = hb.bin_idx(hb.n_applied), target_c = hb.c_count(target), target_i = target_c - hb.n_applied;
    if (target_i >= len) {
      lit_apply_incomplete<K, L>(hb, v, ref, len, true, at(0), true, at(len - 1));
      if (target_i == len) hb.complete_bin(target);
      break;
    }
    const L x = at(target_i - 1);
    uint32_t lo = 0, hi = target_i - 1;   // first index holding x: v[hi] == x
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (at(mid) < x) lo = mid + 1; else hi = mid; }
    const uint32_t l = lo;
    lo = target_i; hi = len;              // first index >= target_i holding something else
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (at(mid) == x) lo = mid + 1; else hi = mid; }
    const uint32_t r = lo;
    if (l > 0) lit_apply_incomplete<K, L>(hb, v, ref, l, true, at(0), true, at(l - 1));
    lit_apply_constant_run<L>(hb, r - l, x);
    v += r; len -= r;
  }
}

// K = uint16_t: the chunk's speculative 16-bit latents are the keys (latent = c16_ref + key; the order is the latents' order), a quarter of the
// bytes a pass moves for 64-bit latents.  Everything the reference compares with a BOUND stays in latent space -- its root bounds are the latent
// type's 0 and MAX, and `tentative > lower bound` decides which side of the pivot is tight -- so pivots go key -> latent -> key.
template <class K, class L>
__device__ void lit_var(const EncWorkspace& ws, uint32_t t, uint32_t var, uint32_t bins_log, uint32_t PCO_GLOBAL* fell_back) {
  EncChunk PCO_GLOBAL* ch = (EncChunk PCO_GLOBAL*)ws.chunks + t;
  EncVar PCO_GLOBAL* ev = &ch->v[var];
  const uint32_t lane = lane_id();
  const uint32_t n_lat = uni(ev->n_lat);
  if (n_lat == 0) return;
  uint8_t PCO_LDS* smem = enc_lds_base();
  uint32_t PCO_LDS* stk32 = (uint32_t PCO_LDS*)(smem + kLitLdsStack);
  uint32_t PCO_LDS* cnt = (uint32_t PCO_LDS*)(smem + kLitLdsCnt);
  K PCO_GLOBAL* A = sort_ptr<K>(ws, t, 0);
  K PCO_GLOBAL* T = sort_ptr<K>(ws, t, 1);
  constexpr bool kKeys16 = !std::is_same<K, L>::value;
  const L ref = kKeys16 ? (L)uni((uint64_t)ch->c16_ref[var == 2 ? 1 : 0]) : (L)0;   // latent = ref + key
  {  // the stored latents in order (collect_contiguous_latents, wrapped/chunk_compressor.rs:128-140): every position that is not among the first `skip` of its page
    const L PCO_GLOBAL* lat = lat_ptr<L>(ws, t, var);
    const uint16_t PCO_GLOBAL* clat = clat_ptr(ws, t, var);
    const bool c16 = uni(ch->c16_ok) == 1 && var != 0;
    const L cref = (L)uni((uint64_t)ch->c16_ref[var == 2 ? 1 : 0]);
    const uint32_t n_all = (uint32_t)uni((uint64_t)ch->n), skip = uni(ev->lat_start), plow = uni(ch->page_low), pr = uni(ch->page_r);
    const bool single_page = uni(ch->n_pages) == 1, exact_paging = uni(ch->exact_paging) != 0; const uint32_t n_pg = uni(ch->n_pages);
    const EncPage PCO_GLOBAL* pgl = (const EncPage PCO_GLOBAL*)ws.pages + uni(ch->page_first);
    auto exact_start = [&](uint32_t i) {
      uint32_t lo = 0, hi = n_pg;
      while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if ((uint64_t)pgl[mid].start <= i) lo = mid; else hi = mid; }
      return (uint64_t)pgl[lo].start;
    };
    auto stored = [&](uint32_t i) { return skip == 0 || (single_page ? i >= skip : (uint64_t)i - (exact_paging ? exact_start(i) : page_start_of(i, plow, pr)) >= skip); };
    const uint64_t below = ((uint64_t)1 << lane) - 1;
    uint32_t out = 0;
    for (uint32_t base = 0; base < n_all; base += 64) {
      const uint32_t i = base + lane;
      const bool st = i < n_all && stored(i);
      const uint64_t bal = __ballot(st);
      if (st) A[out + (uint32_t)__popcll(bal & below)] = kKeys16 ? (K)clat[i] : (K)(c16 ? (L)(cref + (L)clat[i]) : lat[i]);
      out += (uint32_t)__popcll(bal);
    }
    lit_sync();
  }
  LitBuilder<L> hb; hb.n = n_lat; hb.bins_log = bins_log; hb.plan = plan_ref(ws, t, var);
  // explicit recursion (histograms.rs:208-280): descend into the left child, keep the right sibling on the stack
  auto push = [&](uint32_t sp, uint32_t lo, uint32_t len, L lbx, bool lbt, L ubx, bool ubt, uint32_t limit) {
    if (lane == 0) {
      stk32[6 * sp] = lo; stk32[6 * sp + 1] = len | (limit << 25) | ((lbt ? 1u : 0u) << 30) | ((ubt ? 1u : 0u) << 31);
      stk32[6 * sp + 2] = (uint32_t)(uint64_t)lbx; stk32[6 * sp + 3] = (uint32_t)((uint64_t)lbx >> 32);
      stk32[6 * sp + 4] = (uint32_t)(uint64_t)ubx; stk32[6 * sp + 5] = (uint32_t)((uint64_t)ubx >> 32);
    }
  };
  uint32_t limit0 = 0; { uint64_t x = (uint64_t)n_lat + 1; while (x >>= 1) limit0++; limit0 += 1; }   // 1 + ilog2(n + 1)  (histograms.rs:35-41)
  uint32_t sp = 0, guard = 0; bool overflow = false, fell = false;   // guard: a replay that does not terminate is a bug, not a hang (every node either ends or shrinks)
  push(sp++, 0u, n_lat, (L)0, false, (L)~(L)0, false, limit0);
  lit_sync();
  while (sp > 0 && !overflow) {
    sp--;
    uint32_t lo = uni(stk32[6 * sp]); const uint32_t w1 = uni(stk32[6 * sp + 1]);
    uint32_t len = w1 & 0x1ffffffu, limit = (w1 >> 25) & 31u; bool lbt = ((w1 >> 30) & 1u) != 0, ubt = (w1 >> 31) != 0;
    L lbx = (L)(((uint64_t)uni(stk32[6 * sp + 3]) << 32) | uni(stk32[6 * sp + 2])), ubx = (L)(((uint64_t)uni(stk32[6 * sp + 5]) << 32) | uni(stk32[6 * sp + 4]));
    lit_sync();
    for (;;) {
      if (len == 0) break;
      if (++guard > 1024u + 4u * n_lat) { overflow = true; break; }
      K PCO_GLOBAL* v = A + lo;
      const uint32_t target = hb.bin_idx(hb.n_applied), target_c = hb.c_count(target), end = hb.n_applied + len;
      if (end <= target_c) {
        lit_apply_incomplete<K, L>(hb, v, ref, len, lbt, lbx, ubt, ubx);
        if (end == target_c) hb.complete_bin(target);
        break;
      }
      if (lbx == ubx || len == 1) { lit_apply_constant_run<L>(hb, len, (L)(ref + (L)lit_uni<K>(v[0]))); break; }
      const L tentative = (L)(ref + (L)lit_choose_pivot<K>(v, len));
      L pivot, lhs_ubx, rhs_lbx; bool lhs_ubt, rhs_lbt;
      if (tentative > lbx) { pivot = tentative; lhs_ubx = (L)(tentative - 1); lhs_ubt = false; rhs_lbx = tentative; rhs_lbt = true; }
      else { pivot = (L)(tentative + 1); lhs_ubx = tentative; lhs_ubt = true; rhs_lbx = (L)(tentative + 1); rhs_lbt = false; }
      static_assert(kC16KeyRange <= (1u << 15), "the 16-bit replay carries pivot - ref = key + 1 in a uint16_t: keys must stay below 2^15 (enc_split_kernel<c16>)");
      const uint32_t lhs = lit_partition<K>(v, len, (K)(pivot - ref));   // (pivot - ref <= kC16KeyRange = 2^15: tentative is a key below it)
      const uint32_t smaller = lhs < len - lhs ? lhs : len - lhs;
      if (1 + smaller < len / 8) {   // was_bad_pivot (sort_utils.rs:124)
        limit -= 1;
        if (limit == 0) {
          fell = true;
          lit_sort<K>(v, T + lo, len, cnt);
          lit_apply_sorted<K, L>(hb, v, ref, len);
          break;
        }
        lit_break_patterns<K>(v, lhs);
        lit_break_patterns<K>(v + lhs, len - lhs);
      }
      if (sp >= kLitStackCap) { overflow = true; break; }
      push(sp++, lo + lhs, len - lhs, rhs_lbx, rhs_lbt, ubx, ubt, limit);
      lit_sync();
      len = lhs; ubx = lhs_ubx; ubt = lhs_ubt;
    }
  }
  if (lane == 0) {
    if (overflow) ch->status = PCO_GFX_DEVICE_ERROR;   // (cannot happen: the depth bound above)
    ev->n_hist = hb.n_hist < hb.plan.cap ? hb.n_hist : hb.plan.cap;
    if (fell && fell_back != nullptr) atomicAdd((uint32_t*)fell_back, 1u);
  }
  lit_sync();
}

__global__ __launch_bounds__(64) void enc_hist_literal_kernel(EncWorkspace ws, uint32_t n_tasks, uint32_t* fell_back) {
  const uint32_t t = blockIdx.x;
  if (t >= n_tasks) return;
  const EncChunk PCO_GLOBAL* ch = (const EncChunk PCO_GLOBAL*)ws.chunks + t;
  if (uni(ch->status) != PCO_GFX_OK) return;
  const int bits = dtype_bits(uni(ch->dtype));
  const uint32_t ubl = uni(ch->unopt_bins_log);
  uint32_t PCO_GLOBAL* fb = (uint32_t PCO_GLOBAL*)fell_back;
  for (uint32_t var = 0; var < 3; var++) {
    if (!uni(ch->v[var].present)) continue;
    const uint32_t bl = var == 2 ? (ubl < 6 ? ubl : 6) : ubl;   // (wrapped/chunk_compressor.rs:238-248)
    const bool keys16 = var != 0 && uni(ch->c16_ok) == 1;   // the split's 16-bit latents held: they are the keys
    if (var == 0) lit_var<uint32_t, uint32_t>(ws, t, var, bl, fb);
    else if (bits == 64) { if (keys16) lit_var<uint16_t, uint64_t>(ws, t, var, bl, fb); else lit_var<uint64_t, uint64_t>(ws, t, var, bl, fb); }
    else if (bits == 32) { if (keys16) lit_var<uint16_t, uint32_t>(ws, t, var, bl, fb); else lit_var<uint32_t, uint32_t>(ws, t, var, bl, fb); }
    else if (bits == 16) lit_var<uint16_t, uint16_t>(ws, t, var, bl, fb);
    else lit_var<uint8_t, uint8_t>(ws, t, var, bl, fb);
  }
}

}  // namespace pcogfx
