// =============================================================================
// pco_oracle_encode.hpp -- ORACLE (test infrastructure), encode side.
// Restates wrapped/chunk_compressor.rs, chunk_latent_compressor.rs,
// compression_table.rs, histograms.rs, sort_utils.rs, bin_optimization.rs,
// delta/{consecutive,lookback}.rs (encode), mode/*.rs (split + auto detection),
// sampling.rs and standalone/{compressor,simple,guarantee}.rs of pco v1.0.3.
// See pco_oracle.hpp for the rules that govern this code.
// =============================================================================
#pragma once
#include "pco_oracle.hpp"

namespace pco_oracle {

// ----------------------------------------------------------------------------
// config (chunk_config.rs:13-125,191-235)
// ----------------------------------------------------------------------------
enum ModeSpecKind { kModeAuto = 0, kModeClassic = 1, kModeTryFloatMult = 2, kModeTryFloatQuant = 3, kModeTryIntMult = 4, kModeTryDict = 5 };
enum DeltaSpecKind { kDeltaAuto = 0, kDeltaSpecNoOp = 1, kDeltaTryConsecutive = 2, kDeltaTryLookback = 3, kDeltaTryConv1 = 4 };
struct ChunkConfig {
  size_t compression_level = DEFAULT_COMPRESSION_LEVEL;
  ModeSpecKind mode_kind = kModeAuto; double mode_f64 = 0.0; uint64_t mode_u64 = 0;
  DeltaSpecKind delta_kind = kDeltaAuto; size_t delta_order = 0;
  bool paging_exact = false; size_t max_page_n = DEFAULT_MAX_PAGE_N; std::vector<size_t> exact_pages;
  bool enable_8_bit = false;
};
// chunk_config.rs:134-182
inline std::vector<size_t> n_per_page(const ChunkConfig& c, size_t n) {
  std::vector<size_t> res;
  if (!c.paging_exact) {
    if (n == 0) return res;
    if (c.max_page_n == 0) fail(kInvalidArgument, "max page n must be positive");
    size_t n_pages = (n + c.max_page_n - 1) / c.max_page_n;
    size_t low = n / n_pages, r = n % n_pages;
    res.assign(n_pages, low);
    for (size_t i = 0; i < r; i++) res[i] = low + 1;
  } else res = c.exact_pages;
  size_t sum = 0; for (size_t x : res) sum += x;
  if (sum != n) fail(kInvalidArgument, "paging spec suggests a different count of numbers than given");
  for (size_t x : res) if (x == 0) fail(kInvalidArgument, "cannot write data page of 0 numbers");
  return res;
}
// chunk_config.rs:269-314
inline void validate_config(const ChunkConfig& c, int latent_bits) {
  if (c.compression_level > MAX_COMPRESSION_LEVEL) fail(kInvalidArgument, "compression level may not exceed 12");
  if (c.delta_kind == kDeltaTryConsecutive && c.delta_order > MAX_CONSECUTIVE_DELTA_ORDER) fail(kInvalidArgument, "consecutive delta order may not exceed 7");
  if (c.delta_kind == kDeltaTryConv1) {
    if (c.delta_order > MAX_CONV1_DELTA_ORDER) fail(kInvalidArgument, "conv1 delta order may not exceed 32");
    if (latent_bits > 32) fail(kInvalidArgument, "Conv1 delta encoding is only supported for types with 32 or fewer bits");
  }
  if (latent_bits == 8 && !c.enable_8_bit) fail(kInvalidArgument, "compressing 8-bit types with Pco is often a mistake");
}

// ----------------------------------------------------------------------------
// histograms (histograms.rs, sort_utils.rs) -- the LITERAL algorithm, including
// the pivot choice and the heapsort fallback, because the fallback path
// (apply_sorted) can differ from the quickselect path on ties.
// ----------------------------------------------------------------------------
template <class L> struct HistogramBin { size_t count; L lower; L upper; };

namespace sortu {
template <class L> L choose_pivot(const L* v, size_t len) {  // sort_utils.rs:5-56
  size_t a = len / 4, b = len / 2, c = (len * 3) / 4;
  if (len >= 8) {
    auto sort2 = [&](size_t& x, size_t& y) { if (v[y] < v[x]) std::swap(x, y); };
    auto sort3 = [&](size_t& x, size_t& y, size_t& z) { sort2(x, y); sort2(y, z); sort2(x, y); };
    if (len >= 50) {
      auto sort_adjacent = [&](size_t& x) { size_t tmp = x; size_t lo = tmp - 1, hi = tmp + 1; sort3(lo, x, hi); };
      sort_adjacent(a); sort_adjacent(b); sort_adjacent(c);
    }
    sort3(a, b, c);
  }
  return v[b];
}
template <class L> void break_patterns(L* v, size_t len) {  // sort_utils.rs:61-105 (64-bit usize)
  if (len >= 8) {
    uint64_t seed = len;
    auto gen = [&]() { uint64_t r = seed; r ^= r << 13; r ^= r >> 7; r ^= r << 17; seed = r; return seed; };
    uint64_t modulus = 1; while (modulus < len) modulus <<= 1;  // next_power_of_two
    size_t pos = len / 4 * 2;
    for (size_t i = 0; i < 3; i++) {
      uint64_t other = gen() & (modulus - 1);
      if (other >= len) other -= len;
      std::swap(v[pos - 1 + i], v[other]);
    }
  }
}
template <class L> std::pair<size_t, bool> partition(L* v, size_t len, L pivot) {  // sort_utils.rs:109-126
  size_t left = 0;
  for (size_t pos = 0; pos < len; pos++) {
    L value = v[pos]; bool lt = value < pivot;
    v[pos] = v[left]; v[left] = value; left += lt ? 1 : 0;
  }
  bool bad = 1 + std::min(left, len - left) < len / 8;
  return {left, bad};
}
template <class L> void heapsort(L* x, size_t len) {  // sort_utils.rs:130-169
  auto sift_down = [&](size_t n, size_t node) {
    for (;;) {
      size_t child = 2 * node + 1;
      if (child >= n) break;
      if (child + 1 < n) child += (x[child] < x[child + 1]) ? 1 : 0;
      if (x[node] >= x[child]) break;
      std::swap(x[node], x[child]); node = child;
    }
  };
  for (size_t i = len / 2; i-- > 0;) sift_down(len, i);
  for (size_t i = len; i-- > 1;) { std::swap(x[0], x[i]); sift_down(i, 0); }
}
}  // namespace sortu

template <class L> struct HistogramBuilder {  // histograms.rs:60-281
  struct Bound { bool tight; L x; };
  uint64_t n, n_bins; Bitlen n_bins_log;
  size_t n_applied = 0, next_avail_bin_idx = 0;
  bool has_incomplete = false; HistogramBin<L> incomplete{};
  std::vector<HistogramBin<L>> dst;
  bool used_heapsort_fallback = false;
  HistogramBuilder(size_t n_, Bitlen n_bins_log_) : n(n_), n_bins((uint64_t)1 << n_bins_log_), n_bins_log(n_bins_log_) {}

  void apply_incomplete(const L* v, size_t len, Bound lower, Bound upper) {
    if (len == 0) return;
    auto smax = [&]() { L m = 0; for (size_t i = 0; i < len; i++) m = std::max(m, v[i]); return m; };
    auto smin = [&]() { L m = LMAX<L>(); for (size_t i = 0; i < len; i++) m = std::min(m, v[i]); return m; };
    if (has_incomplete) {
      incomplete.upper = upper.tight ? upper.x : smax();
      incomplete.count += len;
    } else {
      L lb = lower.tight ? lower.x : smin();
      L ub = upper.tight ? upper.x : smax();
      incomplete = HistogramBin<L>{len, lb, ub}; has_incomplete = true;
    }
    n_applied += len;
  }
  bool complete_bin(size_t bin_idx) {
    if (has_incomplete) { next_avail_bin_idx = bin_idx + 1; dst.push_back(incomplete); has_incomplete = false; return true; }
    return false;
  }
  size_t bin_idx(size_t c_count) const { return (size_t)((((uint64_t)c_count) << n_bins_log) / n); }
  size_t c_count(size_t bin_idx_) const { return (size_t)((((uint64_t)bin_idx_ + 1) * n + n_bins - 1) >> n_bins_log); }
  void apply_constant_run(const L* v, size_t len) {
    size_t start = n_applied, mid = start + len / 2, end = start + len;
    size_t b = bin_idx(mid);
    if (b > next_avail_bin_idx) {
      size_t spare = b - 1;
      if (!complete_bin(spare)) b = spare;
    }
    Bound cb{true, v[0]};
    apply_incomplete(v, len, cb, cb);
    if (end >= c_count(b)) complete_bin(b);
  }
  void apply_sorted(const L* v, size_t len) {  // histograms.rs:164-206
    while (len > 0) {
      size_t target_bin_idx = bin_idx(n_applied);
      size_t target_c_count = c_count(target_bin_idx);
      size_t target_i = target_c_count - n_applied;
      if (target_i >= len) {
        apply_incomplete(v, len, Bound{true, v[0]}, Bound{true, v[len - 1]});
        if (target_i == len) complete_bin(target_bin_idx);
        break;
      }
      size_t l = target_i - 1, r = target_i; L target_x = v[l];
      while (l > 0 && v[l - 1] == target_x) l--;
      while (r < len && v[r] == target_x) r++;
      if (l > 0) apply_incomplete(v, l, Bound{true, v[0]}, Bound{true, v[l - 1]});
      apply_constant_run(v + l, r - l);
      v += r; len -= r;
    }
  }
  void apply_quicksort_recurse(L* v, size_t len, Bound lb, Bound ub, uint32_t bad_pivot_limit) {  // :208-280
    if (len == 0) return;
    size_t target_bin_idx = bin_idx(n_applied);
    size_t target_c_count = c_count(target_bin_idx);
    size_t end = n_applied + len;
    if (end <= target_c_count) {
      apply_incomplete(v, len, lb, ub);
      if (end == target_c_count) complete_bin(target_bin_idx);
      return;
    }
    L loose_lb = lb.x;
    if (loose_lb == ub.x || len == 1) { apply_constant_run(v, len); return; }
    L tentative = sortu::choose_pivot(v, len);
    L pivot; Bound lhs_ub, rhs_lb;
    if (tentative > loose_lb) { pivot = tentative; lhs_ub = Bound{false, (L)(tentative - 1)}; rhs_lb = Bound{true, tentative}; }
    else { pivot = (L)(tentative + 1); lhs_ub = Bound{true, tentative}; rhs_lb = Bound{false, (L)(tentative + 1)}; }
    auto pr = sortu::partition(v, len, pivot);
    size_t lhs_count = pr.first;
    if (pr.second) {
      bad_pivot_limit -= 1;
      if (bad_pivot_limit == 0) {
        used_heapsort_fallback = true;
        sortu::heapsort(v, lhs_count); sortu::heapsort(v + lhs_count, len - lhs_count);
        apply_sorted(v, len);
        return;
      }
      sortu::break_patterns(v, lhs_count); sortu::break_patterns(v + lhs_count, len - lhs_count);
    }
    apply_quicksort_recurse(v, lhs_count, lb, lhs_ub, bad_pivot_limit);
    apply_quicksort_recurse(v + lhs_count, len - lhs_count, rhs_lb, ub, bad_pivot_limit);
  }
};
// histograms.rs:294-298.  Mutates `latents`.  `fallback` (optional) reports whether the
// heapsort fallback ran (the GPU path cannot reproduce that input-ORDER-dependent branch).
template <class L> std::vector<HistogramBin<L>> histogram(L* latents, size_t n, Bitlen n_bins_log, bool* fallback = nullptr) {
  HistogramBuilder<L> hb(n, n_bins_log);
  typename HistogramBuilder<L>::Bound lb{false, 0}, ub{false, LMAX<L>()};
  uint32_t bad_pivot_limit = 1 + ilog2_u64((uint64_t)n + 1);
  hb.apply_quicksort_recurse(latents, n, lb, ub, bad_pivot_limit);
  if (fallback) *fallback = hb.used_heapsort_fallback;
  return hb.dst;
}
// The "multiset rule" the GPU implements: apply the quickselect path's semantics to
// fully sorted data.  Used by tests to verify that rule == literal algorithm whenever
// the heapsort fallback did not trigger.
template <class L> std::vector<HistogramBin<L>> histogram_multiset_rule(const L* sorted, size_t n, Bitlen n_bins_log) {
  HistogramBuilder<L> hb(n, n_bins_log);
  size_t i = 0;
  typedef typename HistogramBuilder<L>::Bound Bound;
  while (i < n) {
    size_t target = hb.bin_idx(hb.n_applied), c = hb.c_count(target);
    // maximal run of equal values starting at i
    size_t j = i + 1; while (j < n && sorted[j] == sorted[i]) j++;
    if (j <= c) {  // run fits inside the current bin -> absorbed
      hb.apply_incomplete(sorted + i, j - i, Bound{true, sorted[i]}, Bound{true, sorted[i]});
      if (j == c) hb.complete_bin(target);
    } else hb.apply_constant_run(sorted + i, j - i);
    i = j;
  }
  return hb.dst;
}

// ----------------------------------------------------------------------------
// bin optimization (bin_optimization.rs)
// ----------------------------------------------------------------------------
template <class L> struct BinCompressionInfo { uint32_t weight; L lower; L upper; Bitlen offset_bits; uint32_t symbol; };

inline float f32_from_bits(uint32_t b) { float f; std::memcpy(&f, &b, 4); return f; }
inline uint32_t f32_to_bits(float f) { uint32_t b; std::memcpy(&b, &f, 4); return b; }
inline float log2_approx(float x) {  // bin_optimization.rs:19-43
  const float Z = 0.674f;
  const uint32_t SIGNIF_MASK = 0x7FFFFF;
  const uint32_t Z_SIGNIF = f32_to_bits(Z) & SIGNIF_MASK;
  const float B = 2.0f / Z;
  const float C = -B / (6.0f * Z);
  const float A = -B - C;
  uint32_t bits = f32_to_bits(x);
  uint32_t exp = bits >> 23;
  uint32_t signif = bits & SIGNIF_MASK;
  uint32_t high_bit = signif > Z_SIGNIF ? 1u : 0u;
  uint32_t log_int = exp + high_bit - 127;
  uint32_t exp2 = 0x7F ^ high_bit;
  float normalized = f32_from_bits((exp2 << 23) | signif);
  volatile float t0 = C * normalized;          // volatile: forbid contraction / reassociation
  volatile float t1 = B + t0;
  volatile float t2 = normalized * t1;
  volatile float t3 = (float)log_int + A;
  return t3 + t2;
}
template <class L> inline float bin_cost(float bin_meta_cost, L lower, L upper, uint32_t count, float total_count_log2) {  // :46-57
  float countf = (float)count;
  volatile float ans_cost = total_count_log2 - log2_approx(countf);
  float offset_cost = (float)bits_to_encode_offset<L>((L)(upper - lower));
  volatile float s = ans_cost + offset_cost;
  volatile float p = s * countf;
  return bin_meta_cost + p;
}
// bin_optimization.rs:104-178 -> vector of (j, i) inclusive
template <class L> std::vector<std::pair<size_t, size_t>> choose_optimized_partitioning(const std::vector<HistogramBin<L>>& bins, Bitlen ans_size_log) {
  size_t nb = bins.size();
  std::vector<uint32_t> c_counts(nb + 1, 0); std::vector<float> best_costs(nb + 1, 0.0f);
  uint32_t c = 0;
  for (size_t i = 0; i < nb; i++) { c += (uint32_t)bins[i].count; c_counts[i + 1] = c; best_costs[i + 1] = std::numeric_limits<float>::quiet_NaN(); }
  uint32_t total_count = c;
  float total_count_log2 = log2_approx((float)c);
  std::vector<size_t> best_js(nb);
  float bin_meta_cost = (float)bin_exact_bit_size(LT<L>::BITS, ans_size_log);
  for (size_t i = 0; i < nb; i++) {
    float best_cost = std::numeric_limits<float>::max(); size_t best_j = (size_t)-1;
    L upper = bins[i].upper; uint32_t c_count_i = c_counts[i + 1];
    for (size_t j = i + 1; j-- > 0;) {
      L lower = bins[j].lower;
      volatile float cost = best_costs[j] + bin_cost<L>(bin_meta_cost, lower, upper, c_count_i - c_counts[j], total_count_log2);
      if (cost < best_cost) { best_cost = cost; best_j = j; }
    }
    best_costs[i + 1] = best_cost; best_js[i] = best_j;
  }
  float best_cost = best_costs[nb];
  float single_bin_cost = bin_cost<L>(bin_meta_cost, bins[0].lower, bins[nb - 1].upper, total_count, total_count_log2);
  {
    volatile float bias = 0.1f * (float)total_count;
    volatile float thr = best_cost + bias;
    if (single_bin_cost < thr) return {{0, nb - 1}};
  }
  bool all_trivial = true;
  for (const auto& b : bins) if (b.lower != b.upper) { all_trivial = false; break; }
  if (all_trivial) {
    volatile float cost = 0.0f;
    for (const auto& b : bins) cost = cost + bin_cost<L>(bin_meta_cost, b.lower, b.upper, (uint32_t)b.count, total_count_log2);
    volatile float bias = 0.1f * (float)total_count;
    volatile float thr = best_cost + bias;
    if (cost < thr) {
      std::vector<std::pair<size_t, size_t>> p; for (size_t i = 0; i < nb; i++) p.push_back({i, i}); return p;
    }
  }
  std::vector<std::pair<size_t, size_t>> part;  // rewind_best_partitioning :81-96
  size_t i = nb - 1;
  for (;;) { size_t j = best_js[i]; part.push_back({j, i}); if (j > 0) i = j - 1; else break; }
  std::reverse(part.begin(), part.end());
  return part;
}
template <class L> std::vector<BinCompressionInfo<L>> optimize_bins(const std::vector<HistogramBin<L>>& bins, Bitlen ans_size_log) {  // :180-198
  auto part = choose_optimized_partitioning<L>(bins, ans_size_log);
  std::vector<BinCompressionInfo<L>> res;
  for (size_t s = 0; s < part.size(); s++) {
    size_t j = part[s].first, i = part[s].second; size_t count = 0;
    for (size_t k = j; k <= i; k++) count += bins[k].count;
    res.push_back(BinCompressionInfo<L>{(uint32_t)count, bins[j].lower, bins[i].upper,
                                        bits_to_encode_offset<L>((L)(bins[i].upper - bins[j].lower)), (uint32_t)s});
  }
  return res;
}

// ----------------------------------------------------------------------------
// train_infos (wrapped/chunk_compressor.rs:38-99)
// ----------------------------------------------------------------------------
template <class L> struct TrainedBins { std::vector<BinCompressionInfo<L>> infos; Bitlen ans_size_log = 0; std::vector<uint32_t> counts; bool hist_fallback = false; };
// TEST HOOK (not the reference's behaviour): 1 = the encoder takes its histograms by the multiset rule, i.e. the quickselect path's result as
// a function of the sorted latents, whatever their order -- what the GPU computes.  On an order that sends the literal algorithm into its
// heapsort branch the two differ on ties; with the hook the tests can say EXACTLY what the GPU's bytes are in that case.
inline int& hist_rule_hook() { static thread_local int rule = 0; return rule; }
template <class L> TrainedBins<L> train_infos(std::vector<L> latents, Bitlen unoptimized_bins_log) {
  TrainedBins<L> t;
  if (latents.empty()) return t;
  size_t n_latents = latents.size();
  std::vector<HistogramBin<L>> unopt;
  if (hist_rule_hook() == 1) { std::sort(latents.begin(), latents.end()); unopt = histogram_multiset_rule<L>(latents.data(), n_latents, unoptimized_bins_log); }
  else unopt = histogram<L>(latents.data(), n_latents, unoptimized_bins_log, &t.hist_fallback);
  Bitlen n_log_ceil = n_latents <= 1 ? 0 : ilog2_u64(n_latents - 1) + 1;
  Bitlen estimated_ans_size_log = std::min(std::min(unoptimized_bins_log + 2, (Bitlen)MAX_COMPRESSION_LEVEL), n_log_ceil);
  t.infos = optimize_bins<L>(unopt, estimated_ans_size_log);
  for (auto& info : t.infos) t.counts.push_back(info.weight);
  auto q = quantize_weights(t.counts, n_latents, estimated_ans_size_log);
  t.ans_size_log = q.first;
  for (size_t i = 0; i < t.infos.size(); i++) t.infos[i].weight = q.second[i];
  return t;
}

// ----------------------------------------------------------------------------
// delta encode (delta/consecutive.rs:3-33, delta/lookback.rs:22-185, delta/mod.rs:37-48)
// ----------------------------------------------------------------------------
template <class L> std::vector<L> consecutive_encode_in_place(size_t order, L* latents, size_t len) {
  std::vector<L> moments;
  for (size_t o = 0; o < order; o++) {
    moments.push_back(len > 0 ? latents[0] : (L)0);
    for (size_t i = len; i-- > 1;) latents[i] = (L)(latents[i] - latents[i - 1]);
    size_t trunc = std::min(len, (size_t)1);
    latents += trunc; len -= trunc;
  }
  for (size_t i = 0; i < len; i++) latents[i] = (L)(latents[i] + MID<L>());
  return moments;
}
inline DeltaEncoding new_lookback(size_t n) {  // delta/mod.rs:37-48
  DeltaEncoding d; d.kind = kDeltaLookback;
  Bitlen b = bits_to_encode_offset<uint32_t>((uint32_t)n - 1);
  d.window_n_log = std::min(std::max(b, (Bitlen)4), (Bitlen)15);
  d.state_n_log = 0; d.secondary_uses_delta = false;
  return d;
}
constexpr size_t PROPOSED_LOOKBACKS = 16, BRUTE_LOOKBACKS = 6, REPEATING_LOOKBACKS = 4;
template <class L> std::vector<uint32_t> choose_lookbacks(Bitlen window_n_log, Bitlen state_n_log, const L* latents, size_t len) {
  size_t state_n = (size_t)1 << state_n_log;
  if (len <= state_n) return {};
  size_t hash_table_n = (size_t)1 << (window_n_log + 1);
  size_t window_n = (size_t)1 << window_n_log;
  if (window_n < PROPOSED_LOOKBACKS) fail(kInvalidArgument, "we do not support tiny windows during compression");
  std::vector<uint32_t> lookback_counts(std::min(window_n, len), 1);
  std::vector<uint32_t> lookbacks(len - state_n);
  std::vector<size_t> idx_hash_table(2 * hash_table_n, 0);
  size_t proposed[PROPOSED_LOOKBACKS];
  for (size_t i = 0; i < PROPOSED_LOOKBACKS; i++) proposed[i] = std::min(i + 1, state_n);
  size_t best_lookback = 1, repeating_lookback_idx = 0;
  const size_t hash_mask = hash_table_n - 1;
  auto hash_fn = [&](uint64_t x) { x = (x ^ (x >> 32)) * 11400714819323197441ull; x = x ^ (x >> 32); return (size_t)x & hash_mask; };
  const Bitlen coarsenesses[2] = {0, 8};
  for (size_t i = state_n; i < len; i++) {
    L l = latents[i];
    size_t new_brute = std::min(i, PROPOSED_LOOKBACKS);
    proposed[new_brute - 1] = new_brute;
    {  // hash_lookup (lookback.rs:22-64)
      size_t proposal_idx = BRUTE_LOOKBACKS + REPEATING_LOOKBACKS, offset = 0;
      for (Bitlen coarseness : coarsenesses) {
        uint64_t bucket = (uint64_t)l >> coarseness;
        uint64_t buckets[3] = {bucket - 1, bucket, bucket + 1};
        size_t hashes[3] = {hash_fn(buckets[0]), hash_fn(buckets[1]), hash_fn(buckets[2])};
        for (size_t h : hashes) {
          size_t lb = i - idx_hash_table[offset + h];
          proposed[proposal_idx] = lb <= window_n ? lb : std::min(proposal_idx, i);
          proposal_idx++;
        }
        idx_hash_table[offset + hashes[1]] = i;
        offset += hash_table_n;
      }
    }
    // find_best_lookback (lookback.rs:67-98): strict '>' keeps the FIRST best
    uint32_t best_goodness = 0; size_t new_best = 0;
    for (size_t p = 0; p < PROPOSED_LOOKBACKS; p++) {
      size_t lookback = proposed[p];
      uint32_t cnt = lookback_counts[lookback - 1]; L other = latents[i - lookback];
      uint32_t lookback_goodness = 32 - clz32(cnt);
      L d1 = (L)(l - other), d2 = (L)(other - l); L delta = std::min(d1, d2);
      uint32_t goodness = lookback_goodness + leading_zeros<L>(delta);
      if (goodness > best_goodness) { best_goodness = goodness; new_best = lookback; }
    }
    if (new_best != best_lookback) repeating_lookback_idx++;
    proposed[BRUTE_LOOKBACKS + repeating_lookback_idx % REPEATING_LOOKBACKS] = new_best;
    best_lookback = new_best;
    lookbacks[i - state_n] = (uint32_t)best_lookback;
    lookback_counts[best_lookback - 1]++;
  }
  return lookbacks;
}
template <class L> std::vector<L> lookback_encode_in_place(Bitlen state_n_log, const uint32_t* lookbacks, L* latents, size_t len) {  // lookback.rs:166-185
  size_t state_n = (size_t)1 << state_n_log, real_state_n = std::min(len, state_n);
  for (size_t i = len; i-- > real_state_n;) latents[i] = (L)(latents[i] - latents[i - lookbacks[i - state_n]]);
  std::vector<L> state(state_n, 0);
  for (size_t i = 0; i < real_state_n; i++) state[state_n - real_state_n + i] = latents[i];
  for (size_t i = 0; i < len; i++) latents[i] = (L)(latents[i] + MID<L>());
  return state;
}

// ----------------------------------------------------------------------------
// per latent var compressor (chunk_latent_compressor.rs, compression_table.rs)
// ----------------------------------------------------------------------------
struct PageVarInfo { std::vector<uint64_t> delta_state; size_t start = 0, end = 0; };
struct DissectedVar {
  std::vector<uint32_t> ans_vals, ans_bits, offset_bits; std::vector<uint64_t> offsets;
  uint32_t ans_final_states[4];
};
template <class L> struct LatentCompressor {
  bool present = false;
  LatentVarMeta meta;
  std::vector<BinCompressionInfo<L>> infos;  // sorted by lower
  std::vector<L> search_lowers; size_t search_size_log = 0;
  AnsEncoder encoder; double avg_bits_per_latent = 0;
  bool is_trivial = true, needs_ans = false; Bitlen max_offset_bits = 0;
  std::vector<L> latents; std::vector<uint32_t> counts; bool hist_fallback = false;

  void init(TrainedBins<L> trained, const LatentVarMeta& m, std::vector<L> lat) {  // chunk_latent_compressor.rs:135-161
    present = true; meta = m; latents = std::move(lat); counts = trained.counts; hist_fallback = trained.hist_fallback;
    needs_ans = m.bins.size() != 1;
    infos = trained.infos;  // compression_table.rs:14-33
    search_size_log = infos.size() <= 1 ? 0 : 1 + ilog2_u64(infos.size() - 1);
    std::sort(infos.begin(), infos.end(), [](const BinCompressionInfo<L>& a, const BinCompressionInfo<L>& b) { return a.lower < b.lower; });
    search_lowers.clear(); for (auto& i : infos) search_lowers.push_back(i.lower);
    while (search_lowers.size() < ((size_t)1 << search_size_log)) search_lowers.push_back(LMAX<L>());
    std::vector<uint32_t> weights; max_offset_bits = 0;
    for (auto& b : m.bins) { weights.push_back(b.weight); max_offset_bits = std::max(max_offset_bits, b.offset_bits); }
    encoder.init(m.ans_size_log, weights, spread_state_symbols(m.ans_size_log, weights));
    // metadata/bins.rs:7-32
    is_trivial = m.bins.empty() || (m.bins.size() == 1 && m.bins[0].offset_bits == 0);
    double total_weight = (double)((uint64_t)1 << m.ans_size_log); avg_bits_per_latent = 0.0;
    for (auto& b : m.bins) {
      double ans_bits = (double)m.ans_size_log - std::log2((double)b.weight);
      avg_bits_per_latent += (ans_bits + (double)b.offset_bits) * (double)b.weight / total_weight;
    }
  }
  // chunk_latent_compressor.rs:194-270 (dissect_page) with compression_table.rs:51-74
  DissectedVar dissect_page(size_t start, size_t end) const {
    DissectedVar d; for (int j = 0; j < 4; j++) d.ans_final_states[j] = encoder.default_state();
    if (is_trivial) return d;
    size_t page_n = end - start;
    d.ans_vals.assign(page_n, 0); d.ans_bits.assign(page_n, 0); d.offset_bits.assign(page_n, 0); d.offsets.assign(page_n, 0);
    size_t n_batches = (page_n + FULL_BATCH_N - 1) / FULL_BATCH_N;
    uint32_t symbols[FULL_BATCH_N];
    for (size_t b = n_batches; b-- > 0;) {
      size_t rs = b * FULL_BATCH_N, re = std::min(rs + FULL_BATCH_N, page_n), bn = re - rs;
      const L* lat = latents.data() + start + rs;
      if (infos.size() <= 1) {
        Bitlen ob = infos.empty() ? 0 : infos[0].offset_bits; L lower = infos.empty() ? (L)0 : infos[0].lower;
        for (size_t i = 0; i < bn; i++) { d.offset_bits[rs + i] = ob; d.offsets[rs + i] = (uint64_t)(L)(lat[i] - lower); symbols[i] = 0; }
      } else {
        for (size_t i = 0; i < bn; i++) {
          size_t idx = 0;
          for (size_t depth = 0; depth < search_size_log; depth++) {
            size_t bis = (size_t)1 << (search_size_log - 1 - depth);
            if (lat[i] >= search_lowers[idx + bis]) idx += bis;
          }
          idx = std::min(idx, infos.size() - 1);
          const auto& info = infos[idx];
          symbols[i] = info.symbol; d.offset_bits[rs + i] = info.offset_bits; d.offsets[rs + i] = (uint64_t)(L)(lat[i] - info.lower);
        }
      }
      // encode_ans_in_reverse (chunk_latent_compressor.rs:96-132): chain j = i mod 4, i descending
      if (encoder.size_log == 0) { for (size_t i = 0; i < bn; i++) d.ans_bits[rs + i] = 0; }
      else for (size_t i = bn; i-- > 0;) {
        size_t j = i % ANS_INTERLEAVING; uint32_t ns; Bitlen bits;
        encoder.encode(d.ans_final_states[j], symbols[i], ns, bits);
        d.ans_vals[rs + i] = d.ans_final_states[j] & ((1u << bits) - 1); d.ans_bits[rs + i] = bits;
        d.ans_final_states[j] = ns;
      }
    }
    return d;
  }
  // chunk_latent_compressor.rs:272-329
  void write_dissected_batch(const DissectedVar& d, size_t batch_start, BitWriter& w) const {
    if (batch_start >= d.offsets.size()) return;
    size_t end = std::min(batch_start + FULL_BATCH_N, d.offsets.size());
    if (needs_ans) for (size_t i = batch_start; i < end; i++) w.write_uint(d.ans_vals[i], d.ans_bits[i]);
    if (max_offset_bits != 0) for (size_t i = batch_start; i < end; i++) w.write_uint(d.offsets[i], d.offset_bits[i]);
  }
};

// ----------------------------------------------------------------------------
// ChunkCompressor (wrapped/chunk_compressor.rs:102-705)
// ----------------------------------------------------------------------------
template <class L> struct SplitLatents { std::vector<L> primary; std::vector<L> secondary; bool has_secondary = false; };

template <class L> struct ChunkCompressor {
  ChunkMeta meta; uint8_t dtype = 0;
  LatentCompressor<uint32_t> dvar; LatentCompressor<L> pvar, svar;
  struct PageInfo { size_t page_n; PageVarInfo v[3]; };
  std::vector<PageInfo> page_infos;

  size_t n_pages() const { return page_infos.size(); }
  size_t meta_size_hint() const { return chunk_meta_max_size(meta, LT<L>::BITS); }
  void write_meta(BitWriter& w) const { write_chunk_meta(meta, LT<L>::BITS, w); }
  // chunk_compressor.rs:603-621
  size_t page_size_hint_inner(size_t page_idx, double over) const {
    const PageInfo& pi = page_infos[page_idx]; size_t body_bit_size = 0;
    auto add = [&](double avg, const PageVarInfo& v) { double nums_bit_size = (double)(v.end - v.start) * avg; body_bit_size += (size_t)std::ceil(nums_bit_size * over); };
    if (dvar.present) add(dvar.avg_bits_per_latent, pi.v[0]);
    add(pvar.avg_bits_per_latent, pi.v[1]);
    if (svar.present) add(svar.avg_bits_per_latent, pi.v[2]);
    return chunk_meta_exact_page_meta_size(meta) + (body_bit_size + 7) / 8;
  }
  size_t page_size_hint(size_t page_idx) const { return page_size_hint_inner(page_idx, 1.2); }
  // chunk_compressor.rs:659-705
  void write_page(size_t page_idx, BitWriter& w) const {
    if (page_idx >= page_infos.size()) fail(kInvalidArgument, "page idx exceeds num pages");
    const PageInfo& pi = page_infos[page_idx];
    DissectedVar dd, dp, ds;
    if (dvar.present) dd = dvar.dissect_page(pi.v[0].start, pi.v[0].end);
    dp = pvar.dissect_page(pi.v[1].start, pi.v[1].end);
    if (svar.present) ds = svar.dissect_page(pi.v[2].start, pi.v[2].end);
    auto write_var_meta = [&](const PageVarInfo& v, const DissectedVar& d, int latent_bits, Bitlen ans_size_log, uint32_t default_state) {
      for (uint64_t x : v.delta_state) w.write_uint(x, (Bitlen)latent_bits);
      for (int j = 0; j < 4; j++) w.write_uint(d.ans_final_states[j] - default_state, ans_size_log);
    };
    if (dvar.present) write_var_meta(pi.v[0], dd, 32, dvar.encoder.size_log, dvar.encoder.default_state());
    write_var_meta(pi.v[1], dp, LT<L>::BITS, pvar.encoder.size_log, pvar.encoder.default_state());
    if (svar.present) write_var_meta(pi.v[2], ds, LT<L>::BITS, svar.encoder.size_log, svar.encoder.default_state());
    w.finish_byte();
    for (size_t batch_start = 0; batch_start < pi.page_n; batch_start += FULL_BATCH_N) {
      if (dvar.present) dvar.write_dissected_batch(dd, batch_start, w);
      pvar.write_dissected_batch(dp, batch_start, w);
      if (svar.present) svar.write_dissected_batch(ds, batch_start, w);
    }
    w.finish_byte();
  }
};

// wrapped/guarantee.rs:11-37
inline ChunkMeta baseline_chunk_meta(int latent_bits) {
  ChunkMeta m; m.vars[kVarPrimary].present = true; m.vars[kVarPrimary].latent_bits = latent_bits; m.vars[kVarPrimary].ans_size_log = 0;
  m.vars[kVarPrimary].bins.push_back(DynBin{1, 0, (Bitlen)latent_bits});
  return m;
}
inline size_t wrapped_chunk_size_guarantee(int latent_bits, size_t n) {
  return chunk_meta_max_size(baseline_chunk_meta(latent_bits), latent_bits) + (n * (size_t)latent_bits + 7) / 8;
}
// standalone/guarantee.rs:11-37
inline size_t standalone_header_size() { return 4 + 1 + (BITS_TO_ENCODE_VARINT_POWER + 64 + BITS_TO_ENCODE_STANDALONE_VERSION + 7) / 8 + 2; }
inline size_t standalone_chunk_size(int latent_bits, size_t n) { return 1 + 3 + wrapped_chunk_size_guarantee(latent_bits, n); }
inline size_t standalone_file_size(int latent_bits, size_t n, const ChunkConfig& c) {
  size_t res = standalone_header_size() + 1;
  for (size_t cn : n_per_page(c, n)) res += standalone_chunk_size(latent_bits, cn);
  return res;
}

template <class L> LatentVarMeta var_meta_from_trained(const TrainedBins<L>& t) {
  LatentVarMeta m; m.present = true; m.latent_bits = LT<L>::BITS; m.ans_size_log = t.ans_size_log;
  for (auto& i : t.infos) m.bins.push_back(DynBin{i.weight, (uint64_t)i.lower, i.offset_bits});
  return m;
}

// delta_encode_and_build_page_infos + new_candidate (chunk_compressor.rs:142-287)
template <class L> void new_candidate(ChunkCompressor<L>& cc, SplitLatents<L> lat, const std::vector<size_t>& pages,
                                      const Mode& mode, const DeltaEncoding& de, Bitlen unoptimized_bins_log, uint8_t dtype) {
  cc = ChunkCompressor<L>(); cc.dtype = dtype;
  std::vector<uint32_t> delta_latents;
  size_t start_idx = 0;
  LatentVarDelta dprim = delta_for_latent_var(de, kVarPrimary), dsec = delta_for_latent_var(de, kVarSecondary);
  for (size_t page_n : pages) {
    size_t end_idx = start_idx + page_n;
    typename ChunkCompressor<L>::PageInfo pi; pi.page_n = page_n;
    std::vector<uint32_t> page_lookbacks;
    if (de.kind == kDeltaLookback) page_lookbacks = choose_lookbacks<L>(de.window_n_log, de.state_n_log, lat.primary.data() + start_idx, page_n);
    auto encode_var = [&](std::vector<L>& v, const LatentVarDelta& d, PageVarInfo& out) {
      std::vector<L> st;
      if (d.kind == kDeltaConsecutive) st = consecutive_encode_in_place<L>(d.order, v.data() + start_idx, page_n);
      else if (d.kind == kDeltaLookback) st = lookback_encode_in_place<L>(d.state_n_log, page_lookbacks.data(), v.data() + start_idx, page_n);
      else if (d.kind == kDeltaConv1) fail(kUnsupported, "conv1 encode is outside the hot-path scope");
      for (L x : st) out.delta_state.push_back((uint64_t)x);
      out.start = std::min(start_idx + d.n_latents_per_state(), end_idx); out.end = end_idx;
    };
    encode_var(lat.primary, dprim, pi.v[1]);
    if (lat.has_secondary) encode_var(lat.secondary, dsec, pi.v[2]);
    if (de.kind == kDeltaLookback) {
      pi.v[0].start = delta_latents.size(); pi.v[0].end = delta_latents.size() + page_lookbacks.size();
      delta_latents.insert(delta_latents.end(), page_lookbacks.begin(), page_lookbacks.end());
    }
    cc.page_infos.push_back(pi);
    start_idx = end_idx;
  }
  auto contiguous = [&](auto& v, int key) {
    typename std::remove_reference<decltype(v)>::type res;
    for (auto& pi : cc.page_infos) res.insert(res.end(), v.begin() + pi.v[key].start, v.begin() + pi.v[key].end);
    return res;
  };
  cc.meta.mode = mode; cc.meta.delta = de;
  if (de.kind == kDeltaLookback) {
    auto t = train_infos<uint32_t>(contiguous(delta_latents, 0), unoptimized_bins_log);
    cc.meta.vars[kVarDelta] = var_meta_from_trained(t);
    cc.dvar.init(t, cc.meta.vars[kVarDelta], std::move(delta_latents));
  }
  {
    auto t = train_infos<L>(contiguous(lat.primary, 1), unoptimized_bins_log);
    cc.meta.vars[kVarPrimary] = var_meta_from_trained(t);
    cc.pvar.init(t, cc.meta.vars[kVarPrimary], std::move(lat.primary));
  }
  if (lat.has_secondary) {
    auto t = train_infos<L>(contiguous(lat.secondary, 2), std::min(unoptimized_bins_log, LIMITED_UNOPTIMIZED_BINS_LOG));
    cc.meta.vars[kVarSecondary] = var_meta_from_trained(t);
    cc.svar.init(t, cc.meta.vars[kVarSecondary], std::move(lat.secondary));
  }
  validate_chunk_meta(cc.meta);
}

// chunk_compressor.rs:362-371
inline Bitlen choose_unoptimized_bins_log(size_t compression_level, size_t n) {
  Bitlen level = (Bitlen)compression_level;
  Bitlen log_n = (Bitlen)std::floor(std::log2((double)n));
  Bitlen fast = log_n >= 4 ? log_n - 4 : 0;
  if (level <= fast) return level;
  return fast + (level >= fast ? level - fast : 0) / 2;
}

// ----------------------------------------------------------------------------
// sampling (sampling.rs) + xoroshiro128++ (rand_xoshiro 0.6.0, un-vendored:
// SplitMix64-seeded xoroshiro128++, restated from the published algorithm;
// pinned by the reference KAT sampling.rs:186-202)
// ----------------------------------------------------------------------------
struct Xoroshiro128PlusPlus {
  uint64_t s0, s1;
  explicit Xoroshiro128PlusPlus(uint64_t seed) {  // seed_from_u64: SplitMix64 fills the 16 seed bytes
    auto splitmix = [&]() { seed += 0x9E3779B97F4A7C15ull; uint64_t z = seed; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
    s0 = splitmix(); s1 = splitmix();
  }
  static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
  uint64_t next_u64() {
    uint64_t r = rotl(s0 + s1, 17) + s0;
    uint64_t t = s1 ^ s0;
    s0 = rotl(s0, 49) ^ t ^ (t << 21);
    s1 = rotl(t, 28);
    return r;
  }
};
constexpr size_t MIN_SAMPLE = 10, SAMPLE_RATIO = 40, DELTA_TARGET_GROUP_N = 200;
inline bool calc_sample_n(size_t n, size_t& out) { if (n >= MIN_SAMPLE) { out = MIN_SAMPLE + (n - MIN_SAMPLE) / SAMPLE_RATIO; return true; } return false; }
// sampling.rs:73-100: returns the chosen indices in draw order
inline bool choose_mode_sample_indices(size_t n, std::vector<size_t>& idxs) {
  size_t target;
  if (!calc_sample_n(n, target)) return false;
  Xoroshiro128PlusPlus rng(0);
  std::vector<uint8_t> visited((n + 7) / 8, 0);
  idxs.clear();
  for (size_t j = n - target; j < n; j++) {
    size_t t = (size_t)(rng.next_u64() % ((uint64_t)j + 1));
    size_t idx = (visited[t / 8] & (1 << (t % 8))) ? j : t;
    visited[idx / 8] |= (uint8_t)(1 << (idx % 8));
    idxs.push_back(idx);
  }
  return true;
}
// sampling.rs:27-60
template <class L> bool choose_delta_sample(const std::vector<L>& primary, std::vector<L>& sample) {
  size_t n = primary.size(), target;
  if (!calc_sample_n(n, target)) return false;
  size_t group_n = std::min(DELTA_TARGET_GROUP_N, n), n_groups = (target + DELTA_TARGET_GROUP_N - 1) / DELTA_TARGET_GROUP_N;
  size_t nominal = n_groups * group_n;
  size_t group_stride = group_n + ((n > nominal ? n - nominal : 0) / (std::max(n_groups, (size_t)2) - 1));
  sample.clear();
  for (size_t i = 0; i < n_groups; i++) { size_t gs = group_stride * i; sample.insert(sample.end(), primary.begin() + gs, primary.begin() + gs + group_n); }
  return true;
}
// sampling.rs:108-138.  The reference sums over HashMap iteration order (unspecified);
// this restatement sums in first-insertion order and documents that as its tie-break.
template <class L, class S, class Fn> double est_bits_saved_per_num(const std::vector<S>& sample, Fn primary_fn) {
  std::unordered_map<uint64_t, size_t> index; std::vector<std::pair<size_t, double>> entries;
  for (const S& x : sample) {
    L primary; double bits_saved; primary_fn(x, primary, bits_saved);
    auto it = index.find((uint64_t)primary);
    if (it == index.end()) { index.emplace((uint64_t)primary, entries.size()); entries.push_back({0, 0.0}); it = index.find((uint64_t)primary); }
    entries[it->second].first += 1; entries[it->second].second += bits_saved;
  }
  size_t infrequent_cutoff = std::max((size_t)1, (size_t)((double)sample.size() / 256.0));
  double sum = 0.0;
  for (auto& e : entries) if (e.first <= infrequent_cutoff) sum += e.second;
  return sum / (double)sample.size();
}

// ----------------------------------------------------------------------------
// int mult detection (mode/int_mult.rs:56-228, mode/mod.rs:7-18)
// ----------------------------------------------------------------------------
inline double single_category_entropy(double p) { return (p == 0.0 || p == 1.0) ? 0.0 : -p * std::log2(p); }
inline double worst_case_categorical_entropy(double concentrated_p, double n_categories_m1) {
  return single_category_entropy(concentrated_p) + n_categories_m1 * single_category_entropy((1.0 - concentrated_p) / n_categories_m1);
}
template <class L> L calc_gcd(L x, L y) { if (x == 0) return y; for (;;) { if (y == 0) return x; x %= y; std::swap(x, y); } }
template <class Fn> bool solve_root_by_false_position(Fn f, double lb, double ub, double& out) {
  const double X_TOLERANCE = 1E-4;
  double flb = f(lb), fub = f(ub);
  if (flb > 0.0 || fub < 0.0) return false;
  while (ub - lb > X_TOLERANCE && fub - flb > 0.0) {
    double lb_prop = 0.001 + 0.998 * fub / (fub - flb);
    double mid = lb_prop * lb + (1.0 - lb_prop) * ub;
    double fmid = f(mid);
    if (fmid < 0.0) { lb = mid; flb = fmid; } else { ub = mid; fub = fmid; }
  }
  out = (lb + ub) / 2.0; return true;
}
inline double powi3(double x) { return x * x * x; }  // f64::powi(3) == x*x*x for llvm.powi lowering
inline bool filter_score_triple_gcd(double gcd, size_t triples_w_gcd_, size_t total_triples_, double& score) {
  const double PI = 3.14159265358979323846264338327950288;
  const double ZETA_OF_2 = PI * PI / 6.0, LCB_RATIO = 1.0;
  double triples_w_gcd = (double)triples_w_gcd_, total_triples = (double)total_triples_;
  double prob_per_triple = triples_w_gcd / total_triples;
  double natural = 1.0 / (ZETA_OF_2 * gcd * gcd);
  double stdev = std::sqrt(natural * (1.0 - natural) / total_triples);
  double z_score = (prob_per_triple - natural) / stdev;
  if (z_score < 3.0) return false;
  double lcb = triples_w_gcd - LCB_RATIO * std::sqrt(triples_w_gcd);
  if (lcb <= 0.0) return false;
  double congruence = std::min(ZETA_OF_2 * lcb / total_triples, 1.0);
  double gcd_m1 = gcd - 1.0, gcd_m1_inv_sq = 1.0 / (gcd_m1 * gcd_m1);
  auto f = [&](double p) { return powi3(p) + powi3(1.0 - p) * gcd_m1_inv_sq - congruence; };
  double lb = 1.0 / gcd, ub = std::cbrt(congruence) + std::numeric_limits<double>::epsilon();
  double concentrated_p;
  if (!solve_root_by_false_position(f, lb, ub, concentrated_p)) return false;
  double worst_case_bits_saved = std::log2(gcd) - worst_case_categorical_entropy(concentrated_p, gcd_m1);
  if (worst_case_bits_saved < MULT_REQUIRED_BITS_SAVED_PER_NUM) return false;
  score = worst_case_bits_saved; return true;
}
// int_mult.rs:98-213.  HashMap order in the reference is unspecified; ties on score are
// broken here by keeping the LAST maximum in order of first appearance (max_by_key = last max).
template <class L> bool choose_candidate_base(const std::vector<L>& sample, L& base, double& score_out) {
  std::vector<L> gcds;
  for (size_t i = 0; i + 3 <= sample.size(); i += 3) {
    L a = sample[i], b = sample[i + 1], c = sample[i + 2];
    if (a > b) std::swap(a, b); if (b > c) std::swap(b, c); if (a > b) std::swap(a, b);
    L g = calc_gcd<L>((L)(b - a), (L)(c - a));
    if (g > 1) gcds.push_back(g);
  }
  size_t total_triples = sample.size() / 3;
  std::unordered_map<uint64_t, size_t> index; std::vector<std::pair<L, size_t>> counts;
  for (L g : gcds) { auto it = index.find((uint64_t)g); if (it == index.end()) { index.emplace((uint64_t)g, counts.size()); counts.push_back({g, 1}); } else counts[it->second].second++; }
  bool found = false; uint64_t best_key = 0;
  for (auto& gc : counts) {
    double s;
    if (!filter_score_triple_gcd((double)(uint64_t)gc.first, gc.second, total_triples, s)) continue;
    uint64_t key = float_to_latent_ordered<uint64_t>(s);
    if (!found || key >= best_key) { found = true; best_key = key; base = gc.first; score_out = s; }
  }
  return found;
}
template <class L> bool int_mult_choose_base(const L* latents_ordered_src, size_t n, NumKind kind, L& base) {  // int_mult.rs:215-228
  std::vector<size_t> idxs;
  if (!choose_mode_sample_indices(n, idxs)) return false;
  std::vector<L> sample; for (size_t i : idxs) sample.push_back(to_latent_ordered<L>(latents_ordered_src[i], kind));
  if (sample.size() < MIN_SAMPLE) return false;
  L cand; double bits_saved_per_adj;
  if (!choose_candidate_base<L>(sample, cand, bits_saved_per_adj)) return false;
  double est = est_bits_saved_per_num<L, L>(sample, [&](L x, L& primary, double& saved) { primary = (L)(x / cand); saved = bits_saved_per_adj; });
  if (est > MULT_REQUIRED_BITS_SAVED_PER_NUM) { base = cand; return true; }
  return false;
}

// ----------------------------------------------------------------------------
// float mult / float quant detection (mode/float_mult.rs:62-374, float_quant.rs:73-149)
// ----------------------------------------------------------------------------
template <class L> struct FloatMultConfig { typename FloatOps<L>::F base, inv_base; };
template <class L> struct FM {
  typedef FloatOps<L> FO; typedef typename FO::F F;
  static F ex2(int p) { return float_exp2<L>(p); }
  static F insignificant_float_to(F x) { int spare = (int)(FO::PRECISION_BITS > 6 ? FO::PRECISION_BITS - 6 : 0); return x * ex2(-spare); }
  static bool is_approx_zero(F small, F big) { return small <= insignificant_float_to(big); }
  static bool is_small_remainder(F rem, F orig) { return rem <= orig * ex2(-16); }
  static bool is_imprecise(F value, F err) { return value <= err * ex2(6); }
  static bool approx_pair_gcd(F greater, F lesser, F& out) {
    if (is_approx_zero(lesser, greater) || lesser == greater) return false;
    F machine_eps = ex2(-(int)FO::PRECISION_BITS);
    F gv = greater, ge = 0, lv = lesser, le = 0;
    for (;;) {
      F prev = gv;
      F ratio = FO::round(gv / lv);
      ge = ge + (ratio * le + gv * machine_eps);   // lhs.err += ratio * rhs.err + lhs.value * machine_eps
      gv = FO::fabs_(gv - ratio * lv);
      if (is_small_remainder(gv, prev) || gv <= ge) { out = lv; return true; }
      if (is_approx_zero(gv, greater) || is_imprecise(gv, ge)) return false;
      std::swap(gv, lv); std::swap(ge, le);
    }
  }
  static FloatMultConfig<L> from_base(F base) { return FloatMultConfig<L>{base, (F)1.0 / base}; }
  static FloatMultConfig<L> from_inv_base(F inv) { return FloatMultConfig<L>{(F)1.0 / inv, inv}; }
  static bool choose_config_by_trailing_zeros(const std::vector<F>& sample, FloatMultConfig<L>& out) {
    const Bitlen precision_bits = FO::PRECISION_BITS;
    auto calc_div = [&](int exponent, uint32_t tz) { return exponent - (int)(precision_bits > tz ? precision_bits - tz : 0); };
    int k = std::numeric_limits<int>::max(); size_t count = 0;
    for (F x : sample) {
      uint32_t tz = float_trailing_zeros<L>(x);
      if (x != (F)0 && tz >= 5) { int kp = calc_div(float_exponent<L>(x), tz); count++; k = std::min(k, kp); }
    }
    size_t required = std::max((size_t)std::ceil((double)sample.size() * 0.5), MIN_SAMPLE);
    if (count < required) return false;
    std::vector<L> int_sample;
    Bitlen lshift = LT<L>::BITS - precision_bits - 1; L explicit_mantissa = MID<L>();
    for (F x : sample) {
      int exponent = float_exponent<L>(x);
      int kp = calc_div(exponent, float_trailing_zeros<L>(x));
      if (kp >= k && exponent < k + (int)LT<L>::BITS) {
        Bitlen rshift = LT<L>::BITS - 1 - (uint32_t)(exponent - k);
        L lsh = (L)((FO::to_bits(x) << lshift) | explicit_mantissa);
        int_sample.push_back((L)(lsh >> rshift));
      }
    }
    if (int_sample.size() >= required) {
      L int_base; double sc;
      if (!choose_candidate_base<L>(int_sample, int_base, sc)) int_base = 1;
      F base = (F)int_base * ex2(k);
      out = from_base(base); return true;
    }
    return false;
  }
  static bool approx_sample_gcd_euclidean(const std::vector<F>& sample, F& out) {
    std::vector<F> gcds;
    for (size_t i = 0; i + 1 < sample.size(); i += 2) {
      F a = sample[i], b = sample[i + 1], g;
      if (approx_pair_gcd(std::max(a, b), std::min(a, b), g)) gcds.push_back(g);
    }
    size_t required = 1 + (size_t)std::ceil((double)sample.size() * 0.001);
    if (gcds.size() < required) return false;
    std::sort(gcds.begin(), gcds.end());
    for (double pct : {0.1, 0.3, 0.5}) {
      F cand = gcds[(size_t)(pct * (double)gcds.size())];
      size_t similar = 0;
      for (F g : gcds) if (FO::fabs_(g - cand) < (F)0.01 * cand) similar++;
      if (similar >= required) { out = cand; return true; }
    }
    return false;
  }
  static F center_sample_base(F base, const std::vector<F>& sample) {
    F inv_base = (F)1.0 / base, tweak_sum = 0, tweak_weight = 0;
    for (F x : sample) {
      F mult = FO::round(x * inv_base);
      Bitlen mult_exponent = (Bitlen)float_exponent<L>(mult);
      if (mult_exponent < FO::PRECISION_BITS && mult != (F)0) {
        F overshoot = (mult * base) - x;
        F weight = (F)(double)(FO::PRECISION_BITS - mult_exponent);
        tweak_sum += weight * (overshoot / mult);
        tweak_weight += weight;
      }
    }
    return base - tweak_sum / tweak_weight;
  }
  static FloatMultConfig<L> snap_to_int_reciprocal(F base) {
    F inv_base = (F)1.0 / base, round_inv = FO::round(inv_base);
    F decimal_inv = (F)std::pow(10.0, std::round(std::log10((double)inv_base)));
    if (FO::fabs_(inv_base - round_inv) < (F)0.02) return from_inv_base(round_inv);
    if (FO::fabs_(inv_base - decimal_inv) / inv_base < (F)0.01) return from_inv_base(decimal_inv);
    return from_base(base);
  }
  static bool bits_saved_over_classic(const FloatMultConfig<L>& cfg, const std::vector<F>& sample, double& out) {
    double v = est_bits_saved_per_num<L, F>(sample, [&](F x, L& primary, double& saved) {
      F mult = FO::round(x * cfg.inv_base);
      primary = int_float_to_latent<L>(mult);
      Bitlen me = (Bitlen)float_exponent<L>(mult);
      Bitlen inter_base_bits = FO::PRECISION_BITS > me ? FO::PRECISION_BITS - me : 0;
      L approx = float_to_latent_ordered<L>(mult * cfg.base), xu = float_to_latent_ordered<L>(x);
      L abs_adj = (L)(std::max(xu, approx) - std::min(xu, approx));
      Bitlen adj_bits = 1 + 2 * (LT<L>::BITS - leading_zeros<L>(abs_adj));
      saved = (double)inter_base_bits - (double)adj_bits;
    });
    if (v >= MULT_REQUIRED_BITS_SAVED_PER_NUM) { out = v; return true; }
    return false;
  }
  // float_mult.rs:338-358 (max_by = last maximum)
  static bool compute_bid(const std::vector<F>& sample, FloatMultConfig<L>& cfg, double& bits_saved) {
    FloatMultConfig<L> cands[2]; bool ok[2];
    ok[0] = choose_config_by_trailing_zeros(sample, cands[0]);
    F g; ok[1] = approx_sample_gcd_euclidean(sample, g);
    if (ok[1]) cands[1] = snap_to_int_reciprocal(center_sample_base(g, sample));
    bool found = false;
    for (int i = 0; i < 2; i++) if (ok[i]) {
      double v;
      if (!bits_saved_over_classic(cands[i], sample, v)) continue;
      if (!found || !(v < bits_saved)) { found = true; cfg = cands[i]; bits_saved = v; }  // total_cmp, last max
    }
    return found;
  }
  // float_quant.rs:73-149
  static bool quant_compute_bid(const std::vector<F>& sample, Bitlen& k_out, double& bits_saved_out) {
    std::vector<uint32_t> hist(FO::PRECISION_BITS + 1, 0);
    for (F x : sample) hist[std::min((uint32_t)FO::PRECISION_BITS, float_trailing_zeros<L>(x))]++;
    uint32_t rev = 0; for (size_t i = hist.size(); i-- > 0;) { rev += hist[i]; hist[i] = rev; }
    double sample_len = (double)sample.size(); Bitlen best_k = 0; double best = 0.0;
    for (size_t k = 1; k < hist.size(); k++) {
      if (hist[k] == 0) continue;
      double freq = (double)hist[k] / sample_len;
      uint64_t n_categories = ((uint64_t)1 << k) - 1;
      double saved = (double)k - worst_case_categorical_entropy(freq, (double)n_categories);
      if (saved > best) { best_k = (Bitlen)k; best = saved; } else break;
    }
    Bitlen k = best_k; double per = best;
    double v = est_bits_saved_per_num<L, F>(sample, [&](F x, L& primary, double& saved) { primary = (L)(FO::to_bits(x) >> k); saved = per; });
    if (v > QUANT_REQUIRED_BITS_SAVED_PER_NUM) { k_out = k; bits_saved_out = v; return true; }
    return false;
  }
};

// ----------------------------------------------------------------------------
// mode splits (mode/classic.rs:6-12, int_mult.rs:20-35, float_mult.rs:38-60, float_quant.rs:41-71)
// ----------------------------------------------------------------------------
template <class L> SplitLatents<L> split_classic(const L* bits, size_t n, NumKind kind) {
  SplitLatents<L> s; s.primary.resize(n);
  for (size_t i = 0; i < n; i++) s.primary[i] = to_latent_ordered<L>(bits[i], kind);
  return s;
}
template <class L> SplitLatents<L> split_int_mult(const L* bits, size_t n, NumKind kind, L base) {
  SplitLatents<L> s; s.has_secondary = true; s.primary.resize(n); s.secondary.resize(n);
  for (size_t i = 0; i < n; i++) { L u = to_latent_ordered<L>(bits[i], kind); s.primary[i] = (L)(u / base); s.secondary[i] = (L)(u % base); }
  return s;
}
template <class L> SplitLatents<L> split_float_mult(const L* bits, size_t n, FloatMultConfig<L> cfg) {
  typedef FloatOps<L> FO; typedef typename FO::F F;
  SplitLatents<L> s; s.has_secondary = true; s.primary.resize(n); s.secondary.resize(n);
  for (size_t i = 0; i < n; i++) {
    F num = FO::from_bits(bits[i]);
    F mult = FO::round(num * cfg.inv_base);
    s.primary[i] = int_float_to_latent<L>(mult);
    s.secondary[i] = (L)(to_latent_ordered<L>(bits[i], kFloat) - float_to_latent_ordered<L>(mult * cfg.base) + MID<L>());
  }
  return s;
}
template <class L> SplitLatents<L> split_float_quant(const L* bits, size_t n, Bitlen k) {
  SplitLatents<L> s; s.has_secondary = true; s.primary.resize(n); s.secondary.resize(n);
  L lowest_k_bits_max = (L)(((L)1 << k) - 1);
  for (size_t i = 0; i < n; i++) {
    L num_ = to_latent_ordered<L>(bits[i], kFloat);
    s.primary[i] = (L)(num_ >> k);
    L low = (L)(num_ & lowest_k_bits_max);
    bool sign_positive = (bits[i] & MID<L>()) == 0;
    s.secondary[i] = sign_positive ? low : (L)(lowest_k_bits_max - low);
  }
  return s;
}

// choose mode + split (data_types/unsigned.rs:28-47, float.rs:82-132, compression_intermediates.rs:79-84)
template <class L> SplitLatents<L> choose_mode_and_split(const L* bits, size_t n, uint8_t dtype, const ChunkConfig& cfg, Mode& mode) {
  NumKind kind = dtype_kind(dtype);
  mode = Mode();
  if (kind != kFloat) {
    switch (cfg.mode_kind) {
      case kModeAuto: {
        L base;
        if (int_mult_choose_base<L>(bits, n, kind, base)) { mode.kind = kIntMult; mode.base_latent = (uint64_t)base; }
        break;
      }
      case kModeClassic: break;
      case kModeTryIntMult: mode.kind = kIntMult; mode.base_latent = (uint64_t)(L)cfg.mode_u64; break;
      case kModeTryFloatMult: case kModeTryFloatQuant: fail(kInvalidArgument, "unable to use float mode for ints");
      case kModeTryDict: fail(kUnsupported, "dict mode is outside the hot-path scope");
    }
    if (!mode_is_valid(mode, dtype)) fail(kInvalidArgument, "The chosen mode was invalid for the number type");
    if (mode.kind == kIntMult) return split_int_mult<L>(bits, n, kind, (L)mode.base_latent);
    return split_classic<L>(bits, n, kind);
  }
  // floats (f16 through pco_oracle_half.hpp; there is no 8-bit float type)
  if constexpr (LT<L>::BITS >= 16) {
    typedef FloatOps<L> FO; typedef typename FO::F F;
    FloatMultConfig<L> fm_cfg{};
    switch (cfg.mode_kind) {
      case kModeAuto: {
        // bids: classic (0.0), float mult, float quant; max_by keeps the LAST maximum
        double best = 0.0; int winner = 0; Bitlen qk = 0;
        std::vector<size_t> idxs;
        if (choose_mode_sample_indices(n, idxs)) {
          std::vector<F> sample; const F max_for_sampling = FO::max_value() * (F)0.5;
          for (size_t i : idxs) { F x = FO::from_bits(bits[i]); if (float_is_normal<L>(x)) { F a = FO::fabs_(x); if (a <= max_for_sampling) sample.push_back(a); } }
          if (sample.size() >= MIN_SAMPLE) {
            FloatMultConfig<L> c; double v;
            if (FM<L>::compute_bid(sample, c, v) && !(v < best)) { best = v; winner = 1; fm_cfg = c; }
            Bitlen k; double vq;
            if (FM<L>::quant_compute_bid(sample, k, vq) && !(vq < best)) { best = vq; winner = 2; qk = k; }
          }
        }
        if (winner == 1) { mode.kind = kFloatMult; mode.base_latent = (uint64_t)float_to_latent_ordered<L>(fm_cfg.base); }
        else if (winner == 2) { mode.kind = kFloatQuant; mode.k = qk; }
        break;
      }
      case kModeClassic: break;
      case kModeTryFloatMult: {
        F base = (F)cfg.mode_f64; fm_cfg = FloatMultConfig<L>{base, (F)1.0 / base};
        mode.kind = kFloatMult; mode.base_latent = (uint64_t)float_to_latent_ordered<L>(base); break;
      }
      case kModeTryFloatQuant: mode.kind = kFloatQuant; mode.k = (Bitlen)cfg.mode_u64; break;
      case kModeTryIntMult: fail(kInvalidArgument, "unable to use int mult mode on floats");
      case kModeTryDict: fail(kUnsupported, "dict mode is outside the hot-path scope");
    }
    if (!mode_is_valid(mode, dtype)) fail(kInvalidArgument, "The chosen mode was invalid for the number type");
    if (mode.kind == kFloatMult) return split_float_mult<L>(bits, n, fm_cfg);
    if (mode.kind == kFloatQuant) return split_float_quant<L>(bits, n, mode.k);
    return split_classic<L>(bits, n, kFloat);
  } else fail(kInvalidArgument, "no 8-bit float type");
}

// ----------------------------------------------------------------------------
// auto delta (chunk_compressor.rs:289-394)
// ----------------------------------------------------------------------------
template <class L> float calculate_compressed_sample_size(const std::vector<L>& sample, Bitlen ubl, const DeltaEncoding& de, uint8_t dtype) {
  ChunkCompressor<L>* cc = new ChunkCompressor<L>();
  float size;
  try {
    SplitLatents<L> s; s.primary = sample;
    new_candidate<L>(*cc, std::move(s), {sample.size()}, Mode(), de, ubl, dtype);
    size = (float)(cc->meta_size_hint() + cc->page_size_hint_inner(0, 1.0));
  } catch (...) { delete cc; throw; }
  delete cc;
  return size;
}
template <class L> DeltaEncoding choose_auto_delta_encoding(const std::vector<L>& primary, Bitlen ubl, uint8_t dtype) {
  std::vector<L> sample;
  if (!choose_delta_sample<L>(primary, sample)) return DeltaEncoding();
  size_t sample_n = sample.size();
  DeltaEncoding best; float best_cost = calculate_compressed_sample_size<L>(sample, ubl, DeltaEncoding(), dtype);
  float lookback_penalty = 0.25f * (float)sample_n;
  if (best_cost > lookback_penalty) {
    float lookback_cost = calculate_compressed_sample_size<L>(sample, ubl, new_lookback(sample_n), dtype) + lookback_penalty;
    if (lookback_cost < best_cost) { best = new_lookback(primary.size()); best_cost = lookback_cost; }
  }
  for (size_t order = 1; order <= MAX_CONSECUTIVE_DELTA_ORDER; order++) {
    DeltaEncoding e; e.kind = kDeltaConsecutive; e.order = order;
    float cost = calculate_compressed_sample_size<L>(sample, ubl, e, dtype);
    if (cost < best_cost) { best = e; best_cost = cost; } else break;
  }
  return best;
}
template <class L> DeltaEncoding choose_delta_encoding(const SplitLatents<L>& lat, const ChunkConfig& cfg, Bitlen ubl, uint8_t dtype) {
  size_t n = lat.primary.size(); DeltaEncoding d;
  switch (cfg.delta_kind) {
    case kDeltaAuto: return choose_auto_delta_encoding<L>(lat.primary, ubl, dtype);
    case kDeltaSpecNoOp: return d;
    case kDeltaTryConsecutive: if (cfg.delta_order == 0) return d; d.kind = kDeltaConsecutive; d.order = cfg.delta_order; return d;
    case kDeltaTryLookback: return new_lookback(n);
    case kDeltaTryConv1: if (cfg.delta_order == 0) return d; fail(kUnsupported, "conv1 encode is outside the hot-path scope");
  }
  return d;
}

// should_fallback / fallback (chunk_compressor.rs:396-438,502-541)
template <class L> bool should_fallback(const ChunkCompressor<L>& cc, size_t n) {
  if (cc.meta.delta.kind == kDeltaNone && cc.meta.mode.kind == kClassic) return false;
  size_t n_pages = cc.page_infos.size();
  size_t worst_bits = 7 * n_pages;
  auto add = [&](const LatentVarMeta& m, const std::vector<uint32_t>& counts) {
    for (size_t i = 0; i < m.bins.size() && i < counts.size(); i++)
      worst_bits += (size_t)counts[i] * (size_t)(m.bins[i].offset_bits + m.ans_size_log - (31 - clz32(m.bins[i].weight)));
  };
  if (cc.dvar.present) add(cc.meta.vars[0], cc.dvar.counts);
  add(cc.meta.vars[1], cc.pvar.counts);
  if (cc.svar.present) add(cc.meta.vars[2], cc.svar.counts);
  size_t worst = chunk_meta_max_size(cc.meta, LT<L>::BITS) + n_pages * chunk_meta_exact_page_meta_size(cc.meta) + (worst_bits + 7) / 8;
  return worst > wrapped_chunk_size_guarantee(LT<L>::BITS, n);
}
template <class L> void fallback_chunk_compressor(ChunkCompressor<L>& cc, SplitLatents<L> lat, const std::vector<size_t>& pages, uint8_t dtype) {
  cc = ChunkCompressor<L>(); cc.dtype = dtype;
  size_t n = lat.primary.size(), start = 0;
  for (size_t pn : pages) { typename ChunkCompressor<L>::PageInfo pi; pi.page_n = pn; pi.v[1].start = start; pi.v[1].end = start + pn; cc.page_infos.push_back(pi); start += pn; }
  cc.meta = baseline_chunk_meta(LT<L>::BITS);
  TrainedBins<L> t; t.ans_size_log = 0; t.counts = {(uint32_t)n};
  t.infos.push_back(BinCompressionInfo<L>{1, 0, LMAX<L>(), LT<L>::BITS, 0});
  cc.pvar.init(t, cc.meta.vars[kVarPrimary], std::move(lat.primary));
}

// wrapped::ChunkCompressor::new (chunk_compressor.rs:442-500)
template <class L> void chunk_compressor_new(ChunkCompressor<L>& cc, const L* bits, size_t n, uint8_t dtype, const ChunkConfig& cfg) {
  validate_config(cfg, LT<L>::BITS);
  if (n == 0) fail(kInvalidArgument, "cannot compress empty chunk");
  if (n > MAX_ENTRIES) fail(kInvalidArgument, "count may not exceed 2^24 per chunk");
  Mode mode;
  SplitLatents<L> lat = choose_mode_and_split<L>(bits, n, dtype, cfg, mode);
  Bitlen ubl = choose_unoptimized_bins_log(cfg.compression_level, n);
  DeltaEncoding de = choose_delta_encoding<L>(lat, cfg, ubl, dtype);
  std::vector<size_t> pages = n_per_page(cfg, n);
  new_candidate<L>(cc, std::move(lat), pages, mode, de, ubl, dtype);
  if (should_fallback<L>(cc, n)) fallback_chunk_compressor<L>(cc, split_classic<L>(bits, n, dtype_kind(dtype)), pages, dtype);
}

// ----------------------------------------------------------------------------
// standalone framing (standalone/compressor.rs:12-16,85-105,157-203; simple.rs:22-91)
// ----------------------------------------------------------------------------
inline void write_varint(uint64_t n, BitWriter& w) {
  Bitlen power = n == 0 ? 1 : ilog2_u64(n) + 1;
  w.write_uint(power - 1, BITS_TO_ENCODE_VARINT_POWER);
  w.write_uint(n, power);
}
inline void write_standalone_header(BitWriter& w, size_t n_hint, uint8_t uniform_type) {
  w.write_aligned_bytes(MAGIC_HEADER, 4);
  w.write_uint(CURRENT_STANDALONE_VERSION, BITS_TO_ENCODE_STANDALONE_VERSION);
  w.write_aligned_bytes(&uniform_type, 1);
  write_varint(n_hint, w);
  w.finish_byte();
  const uint8_t ver[2] = {FORMAT_MAJOR, FORMAT_MINOR};
  w.write_aligned_bytes(ver, 2);
}
inline size_t standalone_header_len(size_t n_hint) { BitWriter w; write_standalone_header(w, n_hint, 0); return w.byte_len(); }
template <class L> void write_standalone_chunk(const ChunkCompressor<L>& cc, BitWriter& w) {
  w.write_aligned_bytes(&cc.dtype, 1);
  w.write_uint(cc.page_infos[0].page_n - 1, BITS_TO_ENCODE_N_ENTRIES);
  cc.write_meta(w);
  cc.write_page(0, w);
}
// simple_compress (uniform_type=false) / simple_compress_into (uniform_type=true)
template <class L> std::vector<uint8_t> simple_compress_t(const L* bits, size_t n, uint8_t dtype, const ChunkConfig& cfg, bool uniform_type) {
  BitWriter w;
  write_standalone_header(w, n, uniform_type ? dtype : 0);
  std::vector<size_t> chunks = n_per_page(cfg, n);
  size_t start = 0;
  for (size_t cn : chunks) {
    ChunkConfig c2 = cfg; c2.paging_exact = true; c2.exact_pages = {cn};
    ChunkCompressor<L>* cc = new ChunkCompressor<L>();
    try { chunk_compressor_new<L>(*cc, bits + start, cn, dtype, c2); write_standalone_chunk<L>(*cc, w); }
    catch (...) { delete cc; throw; }
    delete cc;
    start += cn;
  }
  const uint8_t term = MAGIC_TERMINATION_BYTE;
  w.write_aligned_bytes(&term, 1);
  w.buf.resize(w.byte_len());
  return w.buf;
}

}  // namespace pco_oracle
