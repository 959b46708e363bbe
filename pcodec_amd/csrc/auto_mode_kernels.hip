// auto_mode_kernels.hip -- ModeSpec::Auto on the device (included by pco_gfx.hip after encode_kernels.hip).
//
// The reference decides the mode of a chunk from a Floyd sample of 10 + (n - 10) / 40 numbers (sampling.rs:73-100;
// data_types/unsigned.rs:28-47, float.rs:82-132).  Nearly all of that work is integer or plain IEEE arithmetic on the sample --
// GCDs of triples, approximate Euclid on pairs, hash-bucket counts -- and runs here, one block per chunk.  What is left to the
// host is the libm part (f64 sqrt / cbrt / log2 / log10 / pow on O(1) numbers per chunk: mode/int_mult.rs:130-185,
// mode/float_mult.rs:261-275, mode/float_quant.rs:101-118), because only the host's libm reproduces the reference's bits.
// Two kernels with one host step in between:
//   stage 1  auto_int_gcd_kernel / auto_float_stats_kernel : the statistics every bid starts from, and for floats the sample itself
//            (kept numbers, |x|, in order) into a scratch buffer for stage 2;
//   host     scores the GCD lists, snaps the Euclid base, picks the float-quant k: this fixes the candidate configurations;
//   stage 2  auto_saved_kernel : est_bits_saved_per_num (sampling.rs:108-138) of every candidate -- bucket the sample by the
//            candidate's key, keep the buckets of at most n / 256 numbers, add up what they save, in the reference's order where
//            the sum is not exact.
// The host divides, compares with the reference's thresholds and takes the winning bid.  Every IEEE operation here is the one
// the reference performs, in its order (division, round and fabs are correctly rounded / exact on gfx950, contraction is off):
// tests/test_gpu_parity.py checks the float statistics against numpy scalars and the decisions against the oracle.
#pragma once

namespace pcogfx {

constexpr uint32_t kAutoT = 512;      // threads per block of the float statistics kernel
constexpr uint32_t kAutoCap = 6656;   // largest sample handled here: chunks of up to 2^18 + 3720 numbers (beyond: the host path)
constexpr uint32_t kGcdSlots = 2048, kGcdMaxEntries = 192;
struct IntGcdEntry { uint64_t gcd; uint32_t count, first; };

__device__ __forceinline__ uint64_t gcd_u64(uint64_t a, uint64_t b) {   // binary GCD
  if (a == 0) return b;
  if (b == 0) return a;
  const int shift = __builtin_ctzll(a | b);
  a >>= __builtin_ctzll(a);
  do { b >>= __builtin_ctzll(b); if (a > b) { const uint64_t t = a; a = b; b = t; } b -= a; } while (b != 0);
  return a << shift;
}
template <class L> __device__ __forceinline__ uint64_t sorted_triple_gcd(L x, L y, L z) {   // mode/int_mult.rs:98-113
  if (x > y) { const L t = x; x = y; y = t; }
  if (y > z) { const L t = y; y = z; z = t; }
  if (x > y) { const L t = x; x = y; y = t; }
  return gcd_u64((uint64_t)(L)(y - x), (uint64_t)(L)(z - x));
}

// Open-addressing count of the GCDs above 1 (key 0 = empty): u64 keys | u32 counts | u32 index of the first triple with that value
struct GcdTable { uint64_t PCO_LDS* keys; uint32_t PCO_LDS* counts; uint32_t PCO_LDS* firsts; uint32_t PCO_LDS* n_used; uint32_t PCO_LDS* n_out; };
__device__ __forceinline__ void gcd_table_clear(const GcdTable& t) {
  for (uint32_t i = threadIdx.x; i < kGcdSlots; i += blockDim.x) { t.keys[i] = 0; t.counts[i] = 0; t.firsts[i] = 0xffffffffu; }
  if (threadIdx.x == 0) { *t.n_used = 0; *t.n_out = 0; }
}
__device__ __forceinline__ void gcd_table_add(const GcdTable& t, uint64_t v, uint32_t tri) {
  if (v <= 1) return;
  uint32_t slot = (uint32_t)((v * 0x9E3779B97F4A7C15ull) >> 53) & (kGcdSlots - 1);
  for (uint32_t probe = 0; probe < kGcdSlots; probe++, slot = (slot + 1) & (kGcdSlots - 1)) {
    const uint64_t old = atomicCAS((unsigned long long*)&t.keys[slot], 0ull, (unsigned long long)v);
    if (old == 0) atomicAdd((uint32_t*)t.n_used, 1u);
    if (old == 0 || old == v) { atomicAdd((uint32_t*)&t.counts[slot], 1u); atomicMin((uint32_t*)&t.firsts[slot], tri); return; }
  }
}
// A value seen once can never become the candidate (score_gcd: its lower confidence bound w - sqrt(w) is 0), and the long tail of
// large accidental GCDs is all such values: they stay on the device.  Returns through n_entries / overflow (by thread 0).
__device__ __forceinline__ void gcd_table_emit(const GcdTable& t, IntGcdEntry* out, uint32_t* n_entries, uint32_t* overflow) {
  __syncthreads();
  const bool table_full = *t.n_used >= kGcdSlots;
  if (!table_full) for (uint32_t i = threadIdx.x; i < kGcdSlots; i += blockDim.x) if (t.keys[i] != 0 && t.counts[i] > 1) {
    const uint32_t at = atomicAdd((uint32_t*)t.n_out, 1u);
    if (at < kGcdMaxEntries) out[at] = IntGcdEntry{t.keys[i], t.counts[i], t.firsts[i]};
  }
  __syncthreads();
  const bool ovf = table_full || *t.n_out > kGcdMaxEntries;
  if (threadIdx.x == 0) { *n_entries = ovf ? 0u : *t.n_out; *overflow = ovf ? 1u : 0u; }
}

// ---------------------------------------------------------------------------------------------------------------------------
// stage 1, integers (mode/int_mult.rs:56-127): the GCD of every sorted triple of the sample, and how often each value above 1
// occurs.  grid = tasks, 256 threads.
// ---------------------------------------------------------------------------------------------------------------------------
struct IntGcdTask { const void* src; const uint32_t* idx; uint32_t n_idx, dtype; };
struct IntGcdResult { uint32_t n_entries, overflow; uint64_t pad; IntGcdEntry e[kGcdMaxEntries]; };
template <class L> __device__ __forceinline__ uint64_t sample_triple_gcd(const IntGcdTask& g, uint32_t tri, uint32_t num_kind) {
  const L PCO_GLOBAL* src = (const L PCO_GLOBAL*)g.src;
  return sorted_triple_gcd<L>(to_latent_ordered<L>(src[g.idx[3 * tri]], num_kind), to_latent_ordered<L>(src[g.idx[3 * tri + 1]], num_kind), to_latent_ordered<L>(src[g.idx[3 * tri + 2]], num_kind));
}
__global__ __launch_bounds__(256) void auto_int_gcd_kernel(const IntGcdTask* tasks, IntGcdResult* out) {
  __shared__ uint64_t keys[kGcdSlots];
  __shared__ uint32_t counts[kGcdSlots], firsts[kGcdSlots];
  __shared__ uint32_t ctr[2];
  const GcdTable tab{(uint64_t PCO_LDS*)keys, (uint32_t PCO_LDS*)counts, (uint32_t PCO_LDS*)firsts, (uint32_t PCO_LDS*)&ctr[0], (uint32_t PCO_LDS*)&ctr[1]};
  const IntGcdTask g = tasks[blockIdx.x];
  gcd_table_clear(tab);
  __syncthreads();
  const uint32_t num_kind = dtype_kind(g.dtype); const int bits = dtype_bits(g.dtype);
  const uint32_t n_tri = g.n_idx / 3;
  for (uint32_t tri = threadIdx.x; tri < n_tri; tri += 256) {
    const uint64_t v = bits == 64 ? sample_triple_gcd<uint64_t>(g, tri, num_kind) : (bits == 32 ? sample_triple_gcd<uint32_t>(g, tri, num_kind) : (bits == 16 ? sample_triple_gcd<uint16_t>(g, tri, num_kind) : sample_triple_gcd<uint8_t>(g, tri, num_kind)));
    gcd_table_add(tab, v, tri);
  }
  IntGcdResult* r = out + blockIdx.x;
  gcd_table_emit(tab, r->e, &r->n_entries, &r->overflow);
}

// ---------------------------------------------------------------------------------------------------------------------------
// stage 1, floats (data_types/float.rs:82-132).  On s = |x| of the normal, not-too-large numbers of the sample, in sample order:
//   * the trailing-zeros histogram float-quant starts from (mode/float_quant.rs:73-100);
//   * choose_config_by_trailing_zeros (mode/float_mult.rs:145-194): how many have >= 5 trailing zero mantissa bits, the power of two
//     k they are all multiples of, the integers the sample becomes in units of 2^k and those integers' triple-GCD counts;
//   * approx_sample_gcd_euclidean (float_mult.rs:101-142, 197-229): the approximate GCD of every neighbouring pair, sorted; the value
//     at the first of the 10th / 30th / 50th percentile that 1 + ceil(n / 1000) values lie within 1 % of; then center_sample_base
//     (float_mult.rs:239-258) of that value: the per-number terms in parallel, the two running sums by one lane in sample order.
// s goes to the task's scratch buffer for stage 2.  grid = tasks, kAutoT threads, dynamic LDS kAutoStage1LdsBytes.
// ---------------------------------------------------------------------------------------------------------------------------
struct FloatStatsTask { const void* src; const uint32_t* idx; void* sbuf; uint32_t n_idx, dtype; };
struct FloatStatsResult {
  uint32_t s_size, tz5, n_gcd, n_ints;
  uint32_t sim[3], has_euclid;
  int32_t k; uint32_t n_entries, overflow, pad;
  uint64_t base_c;                     // bits of the centred Euclid base (F widened to 64 bits)
  uint32_t hist[56];
  IntGcdEntry e[kGcdMaxEntries];
};
constexpr uint32_t kAutoLdsA = 0, kAutoLdsB = kAutoCap * 8, kAutoLdsKeys = 2 * kAutoCap * 8, kAutoLdsCounts = kAutoLdsKeys + kGcdSlots * 8,
                   kAutoLdsFirsts = kAutoLdsCounts + kGcdSlots * 4, kAutoLdsMisc = kAutoLdsFirsts + kGcdSlots * 4, kAutoStage1LdsBytes = kAutoLdsMisc + 128 * 4;
template <class F> struct FloatScreen;
template <> struct FloatScreen<float> {
  typedef uint32_t L; static constexpr int kPrec = 23, kBias = 127;
  static __device__ __forceinline__ float from_bits(uint32_t b) { return __uint_as_float(b); }
  static __device__ __forceinline__ uint32_t to_bits(float f) { return __float_as_uint(f); }
  static __device__ __forceinline__ float rnd(float x) { return roundf(x); }
};
template <> struct FloatScreen<double> {
  typedef uint64_t L; static constexpr int kPrec = 52, kBias = 1023;
  static __device__ __forceinline__ double from_bits(uint64_t b) { return __longlong_as_double((long long)b); }
  static __device__ __forceinline__ uint64_t to_bits(double f) { return (uint64_t)__double_as_longlong(f); }
  static __device__ __forceinline__ double rnd(double x) { return round(x); }
};
template <class F> __device__ __forceinline__ F screen_pow2(int p) { typedef FloatScreen<F> S; return S::from_bits((typename S::L)((typename S::L)(S::kBias + p) << S::kPrec)); }
template <class F> __device__ __forceinline__ int screen_exponent(F x) { typedef FloatScreen<F> S; typedef typename S::L L; const L m = (L)1 << (sizeof(L) * 8 - 1); return (int)((S::to_bits(x) & (L)~m) >> S::kPrec) - S::kBias; }
template <class F> __device__ __forceinline__ bool screen_pair_gcd(F hi, F lo, F& out) {
  typedef FloatScreen<F> S;
  const F tiny = screen_pow2<F>(-(S::kPrec - 6)), eps = screen_pow2<F>(-S::kPrec), p16 = screen_pow2<F>(-16), p6 = screen_pow2<F>(6);
  if (lo <= hi * tiny || lo == hi) return false;
  F gv = hi, ge = 0, lv = lo, le = 0;
  for (;;) {
    const F prev = gv, ratio = S::rnd(gv / lv);
    ge += ratio * le + gv * eps;
    gv = fabs(gv - ratio * lv);
    if (gv <= prev * p16 || gv <= ge) { out = lv; return true; }
    if (gv <= hi * tiny || gv <= ge * p6) return false;
    const F t = gv; gv = lv; lv = t; const F u = ge; ge = le; le = u;
  }
}
// one value per lane (index i0 + lane) added into a running sum in index order: what a sequential loop over the array would compute
template <class F> __device__ __forceinline__ F seq_add_lanes(F acc, F v, uint64_t valid) {
  if (valid == ~(uint64_t)0) {   // a full row: no per-lane test in the dependent chain
#pragma unroll
    for (int j = 0; j < 64; j++) {
      F vj;
      if constexpr (sizeof(F) == 4) vj = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), j));
      else { const uint64_t b = (uint64_t)__double_as_longlong(v); vj = __longlong_as_double((long long)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), j) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, j))); }
      acc += vj;
    }
    return acc;
  }
#pragma unroll 8
  for (int j = 0; j < 64; j++) {
    if (!((valid >> j) & 1)) continue;
    F vj;
    if constexpr (sizeof(F) == 4) vj = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), j));
    else { const uint64_t b = (uint64_t)__double_as_longlong(v); vj = __longlong_as_double((long long)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), j) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, j))); }
    acc += vj;
  }
  return acc;
}

template <class F> __device__ void float_stage1(const FloatStatsTask& g, FloatStatsResult* r, uint8_t PCO_LDS* smem) {
  typedef FloatScreen<F> S; typedef typename S::L L;
  constexpr uint32_t kBits = sizeof(L) * 8;
  L PCO_LDS* A = (L PCO_LDS*)(smem + kAutoLdsA);                        // s: |x| bits of the kept numbers, in sample order
  L PCO_LDS* B = (L PCO_LDS*)(smem + kAutoLdsB);                        // by turns: the 2^k-unit integers, the pair GCDs, the centring terms
  uint32_t PCO_LDS* misc = (uint32_t PCO_LDS*)(smem + kAutoLdsMisc);    // [0, 56) tz histogram | 56 tz5 | 57 k | 58, 59 table counters | 64.. sims | 68.. per-wave counts
  uint8_t PCO_LDS* wq = smem + kAutoLdsKeys;                            // (after the GCD table is done with) centring weights, one byte per number
  const GcdTable tab{(uint64_t PCO_LDS*)(smem + kAutoLdsKeys), (uint32_t PCO_LDS*)(smem + kAutoLdsCounts), (uint32_t PCO_LDS*)(smem + kAutoLdsFirsts), misc + 58, misc + 59};
  uint32_t PCO_LDS* wcnt = misc + 68;   // u32[NW]
  const uint32_t tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
  constexpr uint32_t T = kAutoT, NW = kAutoT / 64;
  auto wave_total = [&]() { uint32_t t = 0; for (uint32_t w = 0; w < NW; w++) t += wcnt[w]; return t; };
  if (tid < 128) misc[tid] = 0;
  if (tid == 0) ((int PCO_LDS*)misc)[57] = 0x7fffffff;
  gcd_table_clear(tab);
  __syncthreads();
  const L PCO_GLOBAL* src = (const L PCO_GLOBAL*)g.src;
  const L mid = (L)1 << (kBits - 1), exp_all = (L)((1u << (kBits - 1 - S::kPrec)) - 1);
  const L max_bits = (L)(mid - 1 - ((L)1 << S::kPrec)), lim = (L)(max_bits - ((L)1 << S::kPrec));   // MAX_FOR_SAMPLING = half the largest finite value
  auto tz_of = [](L a) { return (uint32_t)__builtin_ctzll((unsigned long long)a | ((unsigned long long)1 << 63)); };
  auto exp_of = [](L a) { return (int)(a >> S::kPrec) - S::kBias; };
  auto div_pow = [](int e, uint32_t tz) { return e - (int)((uint32_t)S::kPrec > tz ? (uint32_t)S::kPrec - tz : 0u); };
  // ---- 1. the sample: order-preserving compaction, 256 at a time ----
  // (all of a thread's sample numbers are requested before the first is used: the gather is two dependent HBM reads per number)
  constexpr uint32_t kRounds = (kAutoCap + T - 1) / T;
  L fetched[kRounds];
#pragma unroll
  for (uint32_t rr = 0; rr < kRounds; rr++) { const uint32_t k = rr * T + tid; fetched[rr] = k < g.n_idx ? src[g.idx[k]] : (L)0; }
  uint32_t kept = 0;
#pragma unroll
  for (uint32_t rr = 0; rr < kRounds; rr++) {
    const uint32_t k0 = rr * T;
    if (k0 >= g.n_idx) break;
    const uint32_t k = k0 + tid;
    L a = 0; bool keep = false;
    if (k < g.n_idx) { a = (L)(fetched[rr] & (L)~mid); const L e = (L)(a >> S::kPrec); keep = e != 0 && e != exp_all && a <= lim; }
    const uint64_t m = __ballot(keep);
    if (lane == 0) wcnt[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t before = kept;
    for (uint32_t w = 0; w < wave; w++) before += wcnt[w];
    if (keep) {
      A[before + (uint32_t)__popcll(m & (((uint64_t)1 << lane) - 1))] = a;
      const uint32_t tz = tz_of(a);
      atomicAdd((uint32_t*)&misc[tz < (uint32_t)S::kPrec ? tz : (uint32_t)S::kPrec], 1u);
      if (tz >= 5) { atomicAdd((uint32_t*)&misc[56], 1u); atomicMin((int*)&misc[57], div_pow(exp_of(a), tz)); }
    }
    kept += wave_total();
    __syncthreads();
  }
  L PCO_GLOBAL* sbuf = (L PCO_GLOBAL*)g.sbuf;
  for (uint32_t i = tid; i < kept; i += T) sbuf[i] = A[i];
  const uint32_t tz5 = misc[56];
  const int kpow = ((int PCO_LDS*)misc)[57];
  const uint32_t required = max((uint32_t)ceil((double)kept * 0.5), 10u), need = 1u + (uint32_t)ceil((double)kept * 0.001);
  // ---- 2. trailing zeros: the sample in units of 2^k, and the GCDs of its triples ----
  uint32_t n_ints = 0;
  if (tz5 >= required) {
    constexpr uint32_t lshift = kBits - S::kPrec - 1;
    for (uint32_t i0 = 0; i0 < kept; i0 += T) {
      const uint32_t i = i0 + tid;
      bool on = false; L v = 0;
      if (i < kept) {
        const L a = A[i]; const int e = exp_of(a);
        if (div_pow(e, tz_of(a)) >= kpow && e < kpow + (int)kBits) { on = true; v = (L)(((L)((L)(a << lshift) | mid)) >> (kBits - 1 - (uint32_t)(e - kpow))); }
      }
      const uint64_t m = __ballot(on);
      if (lane == 0) wcnt[wave] = (uint32_t)__popcll(m);
      __syncthreads();
      uint32_t before = n_ints;
      for (uint32_t w = 0; w < wave; w++) before += wcnt[w];
      if (on) B[before + (uint32_t)__popcll(m & (((uint64_t)1 << lane) - 1))] = v;
      n_ints += wave_total();
      __syncthreads();
    }
    if (n_ints >= required) for (uint32_t tri = tid; tri < n_ints / 3; tri += T) gcd_table_add(tab, sorted_triple_gcd<L>(B[3 * tri], B[3 * tri + 1], B[3 * tri + 2]), tri);
  }
  gcd_table_emit(tab, r->e, &r->n_entries, &r->overflow);   // (barriers inside: B and the table are free afterwards)
  // ---- 3. Euclid: the pairs' approximate GCDs, compacted into B, sorted (positive floats order like their bits) ----
  uint32_t n_g = 0;
  for (uint32_t p0 = 0; 2 * p0 + 1 < kept; p0 += T) {
    const uint32_t pr = p0 + tid;
    F gval = 0; bool ok = false;
    if (2 * pr + 1 < kept) {
      const F x = S::from_bits(A[2 * pr]), y = S::from_bits(A[2 * pr + 1]);
      ok = screen_pair_gcd<F>(x > y ? x : y, x > y ? y : x, gval);
    }
    const uint64_t m = __ballot(ok);
    if (lane == 0) wcnt[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t before = n_g;
    for (uint32_t w = 0; w < wave; w++) before += wcnt[w];
    if (ok) B[before + (uint32_t)__popcll(m & (((uint64_t)1 << lane) - 1))] = S::to_bits(gval);
    n_g += wave_total();
    __syncthreads();
  }
  uint32_t p2 = 2; while (p2 < n_g) p2 <<= 1;
  for (uint32_t i = n_g + tid; i < p2; i += T) B[i] = (L)~(L)0;
  __syncthreads();
  for (uint32_t k = 2; k <= p2; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t i = tid; i < (p2 >> 1); i += T) {
        const uint32_t l = ((i & ~(j - 1)) << 1) | (i & (j - 1)), rr = l | j;
        const L x = B[l], y = B[rr];
        if ((x > y) == ((l & k) == 0)) { B[l] = y; B[rr] = x; }
      }
      __syncthreads();
    }
  }
  F c[3] = {0, 0, 0};
  if (n_g > 0) {
    const double pct[3] = {0.1, 0.3, 0.5};
    for (int q = 0; q < 3; q++) c[q] = S::from_bits(B[(uint32_t)(pct[q] * (double)n_g)]);
    uint32_t sim[3] = {0, 0, 0};
    for (uint32_t i = tid; i < n_g; i += T) {
      const F x = S::from_bits(B[i]);
      for (int q = 0; q < 3; q++) if (fabs(x - c[q]) < (F)0.01 * c[q]) sim[q]++;
    }
    for (int q = 0; q < 3; q++) if (sim[q]) atomicAdd((uint32_t*)&misc[64 + q], sim[q]);
  }
  __syncthreads();
  const uint32_t sim0 = misc[64], sim1 = misc[65], sim2 = misc[66];
  const bool has_euclid = n_g >= need && (sim0 >= need || sim1 >= need || sim2 >= need);
  const F gcd_pick = sim0 >= need ? c[0] : (sim1 >= need ? c[1] : c[2]);
  // ---- 4. centring: base - sum(w * (mult * base - x) / mult) / sum(w) over the numbers whose multiplier has precision to spare ----
  F base_c = 0;
  if (has_euclid) {
    const F base = gcd_pick, inv = (F)1.0 / base;
    for (uint32_t i = tid; i < kept; i += T) {
      const F x = S::from_bits(A[i]);
      const F mult = S::rnd(x * inv);
      const uint32_t me = (uint32_t)screen_exponent<F>(mult);
      uint32_t wbits = 0; F term = 0;
      if (me < (uint32_t)S::kPrec && mult != (F)0) {
        const F over = (mult * base) - x;
        wbits = (uint32_t)S::kPrec - me;
        term = (F)(double)wbits * (over / mult);
      }
      wq[i] = (uint8_t)wbits; B[i] = S::to_bits(term);
    }
    __syncthreads();
    if (wave == 0) {
      // the weights are whole numbers <= 52 and their running sums stay below 2^24: every partial sum is exact in F, so the sum is the
      // integer sum; the terms are added one by one in sample order
      F tsum = 0; uint32_t wsum = 0;
      for (uint32_t i0 = 0; i0 < kept; i0 += 64) {
        const uint32_t i = i0 + lane;
        const uint32_t wb = i < kept ? (uint32_t)wq[i] : 0u;
        const F term = i < kept ? S::from_bits(B[i]) : (F)0;
        tsum = seq_add_lanes<F>(tsum, term, __ballot(i < kept));   // (a number without a term carries +0.0: adding it changes nothing, the running sum is never -0.0)
        wsum += wb;
      }
      const F tw = (F)(double)wave_sum(wsum);
      base_c = base - tsum / tw;
    }
  }
  if (tid < 56) r->hist[tid] = misc[tid];
  if (tid == 0) {
    r->s_size = kept; r->tz5 = tz5; r->n_gcd = n_g; r->n_ints = n_ints; r->sim[0] = sim0; r->sim[1] = sim1; r->sim[2] = sim2;
    r->has_euclid = has_euclid ? 1u : 0u; r->k = kpow; r->pad = 0; r->base_c = (uint64_t)S::to_bits(base_c);
  }
}
__global__ __launch_bounds__(kAutoT) void auto_float_stats_kernel(const FloatStatsTask* tasks, FloatStatsResult* out) {
  const FloatStatsTask g = tasks[blockIdx.x];
  if (dtype_bits(g.dtype) == 64) float_stage1<double>(g, out + blockIdx.x, enc_lds_base());
  else float_stage1<float>(g, out + blockIdx.x, enc_lds_base());
}

// ---------------------------------------------------------------------------------------------------------------------------
// stage 2: est_bits_saved_per_num (sampling.rs:108-138) of one candidate.  Bucket the sample by the candidate's key, keep the
// buckets holding at most max(1, n / 256) numbers, add up what their numbers save.
//   kind 0  float mult (mode/float_mult.rs:277-316): key = int-float latent of round(x * inv_base); a number saves
//           (precision bits the multiplier leaves unused) - (1 + 2 * bit length of its adjustment): integers, summed exactly;
//   kind 1  float quant (mode/float_quant.rs:120-149): key = bits >> k; every number saves the same f64 `saved`;
//   kind 2  int mult (mode/int_mult.rs:205-228): key = latent / base; every number saves the same f64 `saved`.
// With a constant saving the reference adds it up per bucket (count times, from 0.0) and then adds the buckets in the order they
// first appeared: T[count] is tabulated, the rare buckets' counts are compacted in first-appearance order and one lane adds them.
// grid = candidates, 256 threads, dynamic LDS kAutoSavedLdsBytes.
// ---------------------------------------------------------------------------------------------------------------------------
struct SavedTask { const void* src; const uint32_t* idx; uint32_t kind, dtype, n, bk; uint64_t base, inv, divisor; double saved; };
struct SavedResult { long long s_int; double s_dbl; };
constexpr uint32_t kSavedSlots = 8192;
constexpr uint32_t kSavedLdsKeys = 0, kSavedLdsCounts = kSavedSlots * 8, kSavedLdsFirsts = kSavedLdsCounts + kSavedSlots * 4, kSavedLdsList = kSavedLdsFirsts + kSavedSlots * 4,
                   kSavedLdsT = kSavedLdsList + ((kAutoCap * 2 + 15) & ~15u), kSavedLdsMisc = kSavedLdsT + 64 * 8, kAutoSavedLdsBytes = kSavedLdsMisc + 64 * 4;
constexpr uint64_t kSavedEmpty = ~0ull;   // no key takes this value (see the key functions: latents of finite multipliers, shifted positive bits, quotients)

template <class L> __device__ __forceinline__ uint64_t saved_key(const SavedTask& t, uint32_t i, int& saved_int) {
  saved_int = 0;
  if (t.kind == 2) { const L lat = to_latent_ordered<L>(((const L PCO_GLOBAL*)t.src)[t.idx[i]], dtype_kind(t.dtype)); return (uint64_t)(L)(lat / (L)t.divisor); }
  const L xb = ((const L PCO_GLOBAL*)t.src)[i];   // the float sample written by stage 1: |x| bits
  if (t.kind == 1) return (uint64_t)(L)(xb >> t.bk);
  if constexpr (sizeof(L) >= 4) {
    typedef typename FloatOf<L>::F F;
    constexpr int kPrec = FloatOf<L>::kMantDigits - 1; constexpr uint32_t kBits = sizeof(L) * 8;
    const F x = bits_to_float(xb), inv = bits_to_float((L)t.inv), base = bits_to_float((L)t.base);
    const F mult = round_half_away(x * inv);
    const L key = int_float_to_latent<L>(mult);
    const L mb = (L)(float_to_bits(mult) & (L)~lmid<L>());
    const uint32_t me = (uint32_t)((int)(mb >> kPrec) - (kBits == 64 ? 1023 : 127));
    const uint32_t inter = (uint32_t)kPrec > me ? (uint32_t)kPrec - me : 0u;
    const L approx = to_latent_ordered<L>(float_to_bits(mult * base), kFloat), xu = to_latent_ordered<L>(xb, kFloat);
    const L adj = approx > xu ? (L)(approx - xu) : (L)(xu - approx);
    const uint32_t lz = adj == 0 ? kBits : (uint32_t)__builtin_clzll((unsigned long long)adj) - (64 - kBits);
    saved_int = (int)inter - (int)(1 + 2 * (kBits - lz));
    return (uint64_t)key;
  } else return 0;
}

template <class L> __device__ void saved_body(const SavedTask& t, SavedResult* out, uint8_t PCO_LDS* smem) {
  uint64_t PCO_LDS* keys = (uint64_t PCO_LDS*)(smem + kSavedLdsKeys);
  uint32_t PCO_LDS* counts = (uint32_t PCO_LDS*)(smem + kSavedLdsCounts);
  uint32_t PCO_LDS* firsts = (uint32_t PCO_LDS*)(smem + kSavedLdsFirsts);
  uint16_t PCO_LDS* list = (uint16_t PCO_LDS*)(smem + kSavedLdsList);
  double PCO_LDS* T = (double PCO_LDS*)(smem + kSavedLdsT);
  uint32_t PCO_LDS* misc = (uint32_t PCO_LDS*)(smem + kSavedLdsMisc);   // [0, 4) per-wave counts | 8, 9: the integer sum (u64)
  const uint32_t tid = threadIdx.x, lane = lane_id(), wave = tid >> 6, n = t.n;
  for (uint32_t i = tid; i < kSavedSlots; i += 256) { keys[i] = kSavedEmpty; counts[i] = 0; firsts[i] = 0xffffffffu; }
  if (tid < 64) misc[tid] = 0;
  __syncthreads();
  auto slot_of = [&](uint64_t key, bool insert) {
    uint32_t slot = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 51) & (kSavedSlots - 1);
    for (;; slot = (slot + 1) & (kSavedSlots - 1)) {
      if (insert) { const uint64_t old = atomicCAS((unsigned long long*)&keys[slot], (unsigned long long)kSavedEmpty, (unsigned long long)key); if (old == kSavedEmpty || old == key) return slot; }
      else if (keys[slot] == key) return slot;
    }
  };
  for (uint32_t i = tid; i < n; i += 256) {
    int sv; const uint64_t key = saved_key<L>(t, i, sv);
    const uint32_t slot = slot_of(key, true);
    atomicAdd((uint32_t*)&counts[slot], 1u);
    if (t.kind != 0) atomicMin((uint32_t*)&firsts[slot], i);
  }
  __syncthreads();
  const uint32_t cutoff = max(1u, (uint32_t)((double)n / 256.0));
  if (t.kind == 0) {
    long long mine = 0;
    for (uint32_t i = tid; i < n; i += 256) { int sv; const uint64_t key = saved_key<L>(t, i, sv); if (counts[slot_of(key, false)] <= cutoff) mine += sv; }
    mine = wave_sum(mine);
    if (lane == 0) atomicAdd((unsigned long long*)&misc[8], (unsigned long long)mine);
    __syncthreads();
    if (tid == 0) { out->s_int = (long long)*(uint64_t PCO_LDS*)&misc[8]; out->s_dbl = 0.0; }
    return;
  }
  // constant saving: the rare buckets' counts in the order the buckets first appeared
  uint32_t n_list = 0;
  for (uint32_t i0 = 0; i0 < n; i0 += 256) {
    const uint32_t i = i0 + tid;
    bool on = false; uint32_t cnt = 0;
    if (i < n) { int sv; const uint32_t slot = slot_of(saved_key<L>(t, i, sv), false); cnt = counts[slot]; on = firsts[slot] == i && cnt <= cutoff; }
    const uint64_t m = __ballot(on);
    if (lane == 0) misc[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t before = n_list;
    for (uint32_t w = 0; w < wave; w++) before += misc[w];
    if (on) list[before + (uint32_t)__popcll(m & (((uint64_t)1 << lane) - 1))] = (uint16_t)cnt;
    n_list += misc[0] + misc[1] + misc[2] + misc[3];
    __syncthreads();
  }
  if (tid == 0) { double acc = 0.0; T[0] = 0.0; for (uint32_t c = 1; c <= cutoff && c < 64; c++) { acc += t.saved; T[c] = acc; } }
  __syncthreads();
  if (wave == 0) {
    double s = 0.0;
    for (uint32_t j0 = 0; j0 < n_list; j0 += 64) {
      const uint32_t j = j0 + lane;
      const double v = j < n_list ? T[list[j]] : 0.0;
      s = seq_add_lanes<double>(s, v, __ballot(j < n_list));
    }
    if (lane == 0) { out->s_int = 0; out->s_dbl = s; }
  }
}
__global__ __launch_bounds__(256) void auto_saved_kernel(const SavedTask* tasks, SavedResult* out) {
  const SavedTask t = tasks[blockIdx.x];
  const int bits = dtype_bits(t.dtype);
  if (bits == 64) saved_body<uint64_t>(t, out + blockIdx.x, enc_lds_base());
  else if (bits == 32) saved_body<uint32_t>(t, out + blockIdx.x, enc_lds_base());
  else if (bits == 16) saved_body<uint16_t>(t, out + blockIdx.x, enc_lds_base());
  else saved_body<uint8_t>(t, out + blockIdx.x, enc_lds_base());
}

}  // namespace pcogfx
