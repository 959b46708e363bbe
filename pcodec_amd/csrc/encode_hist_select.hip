// encode_hist_select.hip -- the exact quantile histogram (histograms.rs) for variables whose value range is too wide to count
// in LDS (max - min >= kWideHistRange), without sorting the variable.
//
// The histogram only asks for the values (and their runs of equal values) at <= 2 * 2^bins_log ranks, so all it needs is a
// MONOTONE map of the latents onto a few thousand buckets whose populations are small wherever a queried rank falls: count
// the buckets, find the buckets that hold queried ranks, gather only those buckets' latents (each bucket into its own
// window, in any order) and order each window by itself -- windows hold some 32 latents and are sorted by one wave in
// registers.  Round 1 bucketed by the top bits of x - min; that map collapses on the distributions delta'd and lookback'd data
// actually have (a dense core plus far outliers, power-law tails, heavy ties): every queried rank then sits in one or two
// buckets and the "subset" is the whole variable (74 ms per launch on the mixed workload, 13 ms on lookback residuals).
// Here the map is built from the data: a 2048-latent sample is sorted in LDS, its 64 quantiles cut the value range into
// segments of (roughly) equal population, and every segment is cut linearly into 128 buckets:
//     bucket(x) = 128 j + ((x - lo_j) >> sh_j),   j = the segment holding x (6-step search over the 64 lower bounds in LDS)
// A value that makes up more than 1/64 of the sample gets a segment of its own, [v, v + 1); segments narrower than 128 values
// have sh_j = 0, i.e. one bucket per value: such "exact" buckets answer rank queries from the bucket counts alone and are
// never gathered.  What is left are buckets of a few dozen latents; a power-law tail puts at most 1/64 of the variable into
// the first bucket of the last segment, which the block orders in LDS (up to 8192 latents).  Anything beyond that (never
// seen in practice) is handed to round 1's radix-sort kernel, which stays as the fallback (EncVar::hist_path == 2).
//
// Per chunk and variable: two streaming passes over the latents (count, gather); two blocks of 512 threads per CU.  Round 6: data
// that is spread evenly over its range needs none of the above -- 64-128 equal power-of-two segments do (stage (B') below), a latent's
// bucket is one shift, and its id does not have to be kept for the gather pass.
#pragma once

namespace pcogfx {

#ifndef PCO_SEL_CELL_LOG
#define PCO_SEL_CELL_LOG 11
#endif
constexpr uint32_t kSelT = 1024;     // threads of enc_hist_small_kernel (block_radix_sort_inplace's array order is built on 16 waves)
constexpr uint32_t kSelThr = 512;    // threads of enc_hist_select_kernel: two blocks per CU, one streaming while the other sorts / scans / queries
constexpr uint32_t kSelSegs = 128, kSelSubLog = 6, kSelSample = 2048, kSelRegionBytes = 32768, kSelBigCap = 256, kSelCellLog = PCO_SEL_CELL_LOG, kSelCells = 1u << kSelCellLog;
static_assert(kSelSegs << kSelSubLog == kSelBuckets, "segments x sub-buckets");
constexpr uint32_t kSelLdsP = 0;                                                // u32[8192 + 8] bucket counts, then exclusive prefix (hist_emit's scratch at the end)
constexpr uint32_t kSelLdsNeed = kSelLdsP + (kSelBuckets + 8) * 4;              // u32[256] bitmap of the buckets to gather
constexpr uint32_t kSelLdsWpre = kSelLdsNeed + kSelBuckets / 8;                 // u16[256 + 8] marked buckets in the bitmap words before this one
constexpr uint32_t kSelLdsNlOc = kSelLdsWpre + (256 + 8) * 2;                   // u32[1600 + 2] first subset index of each window (the gather pass's fill cursors: see (F))
constexpr uint32_t kSelLdsSeg = (kSelLdsNlOc + (kSelMaxNeeded + 2) * 4 + 15) & ~15u;   // {lower bound, scaling}[128], 16 bytes each for 64-bit latents
constexpr uint32_t kSelLdsLut = kSelLdsSeg + kSelSegs * 16;                     // u16[2048 + 8] value cell -> first | last << 8 segment it meets
constexpr uint32_t kSelLdsBig = kSelLdsLut + (kSelCells + 8) * 2;                     // u32[256 + 4] windows the block orders in LDS; counters
constexpr uint32_t kSelLdsRec = (kSelLdsBig + (kSelBigCap + 4) * 4 + 15) & ~15u;   // 32 KB: the sample, then the waves' private sort areas / the block sort area, then (its first
                                                                                    // 12 KB, laid out like enc_hist_kernel's) the rank records and the block-scan scratch
constexpr uint32_t kSelLdsBytes = kSelLdsRec + kSelRegionBytes;
static_assert(kHistLdsCounts <= kSelRegionBytes && 2 * kSelLdsBytes <= 160 * 1024, "two blocks per CU");

// One compare-exchange stage of the bitonic network over the 64 lanes: partner = lane ^ J inside sorted runs of K
template <uint32_t K, uint32_t J, class L> __device__ __forceinline__ L bitonic_stage(L x) {
  const uint32_t lane = lane_id();
  const L o = xor_lane<J>(x);
  const bool keep_min = ((lane & J) == 0) == ((lane & K) == 0);
  const L mn = x < o ? x : o, mx = x < o ? o : x;
  return keep_min ? mn : mx;
}
template <uint32_t K, uint32_t J, class L, uint32_t kN> __device__ __forceinline__ void bitonic_merge(L (&x)[kN]) {
#pragma unroll
  for (uint32_t g = 0; g < kN; g++) x[g] = bitonic_stage<K, J>(x[g]);
  if constexpr (J > 1) bitonic_merge<K, J / 2>(x);
}
template <uint32_t K, class L, uint32_t kN> __device__ __forceinline__ void bitonic_level(L (&x)[kN]) {
  bitonic_merge<K, K / 2>(x);
  if constexpr (K < 64) bitonic_level<K * 2>(x);
}
// ascending bitonic sort of one latent per lane, kN independent sets at once (the kN exchanges of a stage are issued together);
// the exchanges run on DPP / permlane swaps (xor_lane), not through the LDS crossbar
template <class L, uint32_t kN> __device__ __forceinline__ void wave_sort64_multi(L (&x)[kN]) { bitonic_level<2>(x); }
template <class L> __device__ __forceinline__ L wave_sort64(L x) { L a[1] = {x}; wave_sort64_multi<L, 1>(a); return a[0]; }
// Ascending bitonic sort of the 2048-latent sample by the 512 threads of the block, four keys per thread, element i = wave * 256 +
// r * 64 + lane: the exchanges at distance < 64 run on the lanes (xor_lane), those at distance 64 / 128 between a thread's own keys,
// and only the six at distance >= 256 cross the waves through `ex` (L[2048]), where the sorted sample is left.
template <uint32_t J, class L> __device__ __forceinline__ void sample_lane_stage(L (&key)[4], uint32_t ibase, uint32_t k) {
#pragma unroll
  for (uint32_t r = 0; r < 4; r++) {
    const uint32_t i = ibase + r * 64;
    const L o = xor_lane<J>(key[r]);
    const bool keep_min = ((i & J) == 0) == ((i & k) == 0);
    const L mn = key[r] < o ? key[r] : o, mx = key[r] < o ? o : key[r];
    key[r] = keep_min ? mn : mx;
  }
}
template <class L> __device__ __forceinline__ void block_sort_sample(L (&key)[4], L PCO_LDS* ex) {
  static_assert(kSelSample == 2048 && kSelThr == 512, "four keys per thread");
  const uint32_t ibase = (threadIdx.x >> 6) * 256 + (threadIdx.x & 63);
#pragma unroll
  for (uint32_t k = 2; k <= kSelSample; k <<= 1) {
#pragma unroll
    for (uint32_t j = k >> 1; j >= 256; j >>= 1) {   // across waves
#pragma unroll
      for (uint32_t r = 0; r < 4; r++) ex[ibase + r * 64] = key[r];
      __syncthreads();
#pragma unroll
      for (uint32_t r = 0; r < 4; r++) {
        const uint32_t i = ibase + r * 64;
        const L o = ex[i ^ j];
        const bool keep_min = ((i & j) == 0) == ((i & k) == 0);
        const L mn = key[r] < o ? key[r] : o, mx = key[r] < o ? o : key[r];
        key[r] = keep_min ? mn : mx;
      }
      __syncthreads();
    }
#pragma unroll
    for (uint32_t jr = 2; jr >= 1; jr >>= 1) {       // between the thread's own keys (distance 128, 64)
      if (k < 128 * jr) continue;
#pragma unroll
      for (uint32_t r = 0; r < 4; r++) {
        if (r & jr) continue;
        const bool asc = ((ibase + r * 64) & k) == 0;
        const L a = key[r], b = key[r | jr];
        const L mn = a < b ? a : b, mx = a < b ? b : a;
        key[r] = asc ? mn : mx; key[r | jr] = asc ? mx : mn;
      }
    }
    if (k >= 64) sample_lane_stage<32>(key, ibase, k);
    if (k >= 32) sample_lane_stage<16>(key, ibase, k);
    if (k >= 16) sample_lane_stage<8>(key, ibase, k);
    if (k >= 8) sample_lane_stage<4>(key, ibase, k);
    if (k >= 4) sample_lane_stage<2>(key, ibase, k);
    sample_lane_stage<1>(key, ibase, k);
  }
#pragma unroll
  for (uint32_t r = 0; r < 4; r++) ex[ibase + r * 64] = key[r];
  __syncthreads();
}
// ascending bitonic sort of a[0 .. n) in LDS by the whole block, n a power of two (every thread of the block calls this)
template <class L> __device__ __forceinline__ void block_sort_lds(L PCO_LDS* a, uint32_t n) {
  const uint32_t tid = threadIdx.x;
  for (uint32_t k = 2; k <= n; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t i = tid; i < (n >> 1); i += kSelThr) {
        const uint32_t l = ((i & ~(j - 1)) << 1) | (i & (j - 1)), r = l | j;
        const L x = a[l], y = a[r];
        const bool asc = (l & k) == 0;
        if ((x > y) == asc) { a[l] = y; a[r] = x; }
      }
      __syncthreads();
    }
  }
}

// Ascending LSD radix sort of a[0 .. n), n <= 8192, in place, by the whole block of 1024 threads: 8-bit digits, only the `sig_bits`
// significant bits (the Auto-delta trial samples are a dense core plus a few dozen outliers at the seams of the sampled groups:
// 3-4 passes where the bitonic network paid 91 barrier-separated stages whatever the data).  Array order is p = wave * 512 +
// round * 64 + lane; a pass holds every key in registers between its counting and its scatter phase, so the array is its own
// destination.  The stable rank of a key among its wave's equal digits comes from eight ballots per round (one per digit bit);
// the waves' digit counts are scanned digit-major.  scratch: u16[256 * 17] counts (row stride 17: the 16 waves of one digit
// sit in different banks) | u32[256] digit bases | u32[4].
constexpr uint32_t kRadixScratchBytes = 256 * 17 * 2 + 64;
template <class K> __device__ __forceinline__ void block_radix_sort_inplace(K PCO_LDS* a, uint8_t PCO_LDS* scratch, uint32_t n, uint32_t sig_bits) {
  const uint32_t tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
  uint16_t PCO_LDS* cnt = (uint16_t PCO_LDS*)scratch;
  uint32_t PCO_LDS* wsum = (uint32_t PCO_LDS*)(scratch + 256 * 17 * 2);
  const uint64_t lt = ((uint64_t)1 << lane) - 1;
  for (uint32_t s = 0; s < sig_bits; s += 8) {
    for (uint32_t d = lane; d < 256; d += 64) cnt[d * 17 + w] = 0;
    K key[8]; uint32_t rk[4];   // rank among the wave's keys of the same digit, two per word
#pragma unroll
    for (uint32_t j = 0; j < 8; j++) {
      const uint32_t p = w * 512 + j * 64 + lane; const bool on = p < n;
      key[j] = on ? a[p] : (K)0;
      const uint32_t d = (uint32_t)(key[j] >> s) & 255u;
      uint64_t m = __ballot(on);
#pragma unroll
      for (uint32_t bit = 0; bit < 8; bit++) {
        const uint32_t mine = (d >> bit) & 1u;
        const uint64_t bb = __ballot(mine != 0);
        const uint32_t flip = mine - 1u;                       // all ones when my bit is clear: peers are the lanes NOT in the ballot
        m &= bb ^ (((uint64_t)flip << 32) | flip);
      }
      const uint32_t before = (uint32_t)__popcll(m & lt);
      uint32_t old = 0;
      if (on) { old = cnt[d * 17 + w]; if (before == 0) cnt[d * 17 + w] = (uint16_t)(old + (uint32_t)__popcll(m)); }
      if (j & 1) rk[j >> 1] |= (old + before) << 16; else rk[j >> 1] = old + before;
    }
    __syncthreads();
    // digit-major exclusive scan of the 256 x 16 counts = a block-wide scan in thread order: thread 4 d + q owns digit d, waves 4 q .. 4 q + 3
    const uint32_t ci = (tid >> 2) * 17 + (tid & 3) * 4;
    const uint32_t c0 = cnt[ci], c1 = cnt[ci + 1], c2 = cnt[ci + 2], c3 = cnt[ci + 3];
    const uint32_t mine4 = c0 + c1 + c2 + c3;
    const uint32_t incl = wave_incl_scan(mine4);
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    const uint32_t ws_incl = wave_incl_scan(lane < 16 ? wsum[lane] : 0u);
    uint32_t base = incl - mine4 + (w == 0 ? 0u : shfl_idx(ws_incl, (int)w - 1));
    cnt[ci] = (uint16_t)base; base += c0; cnt[ci + 1] = (uint16_t)base; base += c1; cnt[ci + 2] = (uint16_t)base; base += c2; cnt[ci + 3] = (uint16_t)base;
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < 8; j++) {
      const uint32_t p = w * 512 + j * 64 + lane;
      if (p < n) { const uint32_t d = (uint32_t)(key[j] >> s) & 255u; a[cnt[d * 17 + w] + ((rk[j >> 1] >> ((j & 1) * 16)) & 0xffffu)] = key[j]; }
    }
    __syncthreads();
  }
}

#ifdef PCO_SEL_TIMING
__device__ unsigned long long g_sel_timing[16];
#define SEL_STAMP(idx) do { if (threadIdx.x == 0) { const unsigned long long _n = __builtin_readcyclecounter(); atomicAdd(&g_sel_timing[idx], _n - sel_t0); sel_t0 = _n; } } while (0)
#else
#define SEL_STAMP(idx) do { } while (0)
#endif

// ascending bitonic sort of a[0 .. n) in LDS by ONE wave, n a power of two <= kSelWaveSortCap
template <class L> __device__ __forceinline__ void wave_sort_lds(L PCO_LDS* a, uint32_t n) {
  const uint32_t lane = lane_id();
  for (uint32_t k = 2; k <= n; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t i = lane; i < (n >> 1); i += 64) {
        const uint32_t l = ((i & ~(j - 1)) << 1) | (i & (j - 1)), r = l | j;
        const L x = a[l], y = a[r];
        const bool asc = (l & k) == 0;
        if ((x > y) == asc) { a[l] = y; a[r] = x; }
      }
      enc_wave_sync();
    }
  }
}

template <class L>
__device__ __forceinline__ void select_var(const EncWorkspace& ws, uint32_t t, uint32_t var, uint32_t bins_log) {
  EncChunk PCO_GLOBAL* ch = (EncChunk PCO_GLOBAL*)ws.chunks + t;
  EncVar PCO_GLOBAL* ev = &ch->v[var];
  const PlanRef plan = plan_ref(ws, t, var);
  const uint32_t tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
  const uint32_t n_lat = ev->n_lat;
  if (n_lat == 0) return;
  const L minv = (L)ev->minv, maxv = (L)ev->maxv;
  if ((uint64_t)(L)(maxv - minv) < kWideHistRange) return;   // the LDS-counting kernels own it
  if (n_lat <= kSmallHistCap) return;                            // enc_hist_small_kernel orders the whole variable in LDS
  const L PCO_GLOBAL* lat = lat_ptr<L>(ws, t, var);
  const uint32_t n_all = (uint32_t)ch->n, skip = ev->lat_start, plow = ch->page_low, pr = ch->page_r;
  const bool single_page = ch->n_pages == 1;
  const bool exact_paging = ch->exact_paging != 0; const uint32_t n_pg = ch->n_pages;
  const EncPage PCO_GLOBAL* pgl = (const EncPage PCO_GLOBAL*)ws.pages + ch->page_first;
  auto exact_start = [&](uint32_t i) {
    uint32_t lo = 0, hi = n_pg;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if ((uint64_t)pgl[mid].start <= i) lo = mid; else hi = mid; }
    return (uint64_t)pgl[lo].start;
  };
  auto stored = [&](uint32_t i) { return skip == 0 || (single_page ? i >= skip : (uint64_t)i - (exact_paging ? exact_start(i) : page_start_of(i, plow, pr)) >= skip); };
  uint8_t PCO_LDS* smem = enc_lds_base();
  L PCO_LDS* rv = (L PCO_LDS*)(smem + kSelLdsRec);
  L PCO_LDS* rnext = (L PCO_LDS*)(smem + kSelLdsRec + 2048);
  L PCO_LDS* rpred = (L PCO_LDS*)(smem + kSelLdsRec + 4096);
  L PCO_LDS* rsucc = (L PCO_LDS*)(smem + kSelLdsRec + 6144);
  uint32_t PCO_LDS* rst = (uint32_t PCO_LDS*)(smem + kSelLdsRec + 8192);
  uint32_t PCO_LDS* ren = (uint32_t PCO_LDS*)(smem + kSelLdsRec + 9216);
  uint32_t PCO_LDS* scan = (uint32_t PCO_LDS*)(smem + kSelLdsRec + 10240);   // u32[512]
  uint32_t PCO_LDS* P = (uint32_t PCO_LDS*)(smem + kSelLdsP);
  uint32_t PCO_LDS* need = (uint32_t PCO_LDS*)(smem + kSelLdsNeed);
  uint16_t PCO_LDS* wpre = (uint16_t PCO_LDS*)(smem + kSelLdsWpre);
  uint32_t PCO_LDS* nl_oc = (uint32_t PCO_LDS*)(smem + kSelLdsNlOc);
  // window (list position) of a marked bucket: its rank among the set bits of the bitmap
  auto slot_of = [&](uint32_t k) { return (uint32_t)wpre[k >> 5] + (uint32_t)__popc(need[k >> 5] & ((1u << (k & 31)) - 1u)); };
  struct alignas(sizeof(L) == 8 ? 16 : 8) SegRec { L lo; uint32_t par; };   // lower bound | scaling (multiplier | pre-shift << 24): one LDS read
  SegRec PCO_LDS* seg = (SegRec PCO_LDS*)(smem + kSelLdsSeg);
  uint16_t PCO_LDS* lut = (uint16_t PCO_LDS*)(smem + kSelLdsLut);
  uint16_t PCO_GLOBAL* ids = clat_ptr(ws, t, var);   // bucket of every latent, written by the count pass for the gather pass (the compact-latent area is idle for wide ranges)
  uint32_t PCO_LDS* big = (uint32_t PCO_LDS*)(smem + kSelLdsBig);   // [0, kSelBigCap): slots; [kSelBigCap]: count; [+1]: fail flag; [+2]: n_seg
  L PCO_LDS* srt = (L PCO_LDS*)(smem + kSelLdsRec);
  constexpr uint32_t NB = kSelBuckets;
  constexpr uint32_t kBlockSortCap = kSelRegionBytes / (sizeof(L) < 4 ? 4 : sizeof(L)), kWaveSortCap = kBlockSortCap / (kSelThr / 64);   // 4096 (8192) and 512 (1024) latents of 8 (<= 4) bytes
  static_assert(kSelSample * sizeof(L) <= kSelRegionBytes, "the sample is sorted in the sort area");
  const uint32_t B = 1u << bins_log;
  const uint64_t n64 = n_lat;
  auto c_count = [&](uint32_t b) { return (uint32_t)((((uint64_t)b + 1) * n64 + B - 1) >> bins_log); };
#ifdef PCO_SEL_TIMING
  unsigned long long sel_t0 = __builtin_readcyclecounter();
#endif
  __syncthreads();
  // ---- (A) sample: 2048 evenly spaced positions (positions that are not stored hold defined junk or, for lookback, possibly
  //      nothing at all: clamping into [min, max] makes any value a harmless boundary candidate), sorted by the block ----
  L skey[4];
#pragma unroll
  for (uint32_t r = 0; r < 4; r++) {
    const uint32_t k = wave * 256 + r * 64 + lane;
    const L x = lat[(uint32_t)(((uint64_t)k * n_all) / kSelSample)];
    skey[r] = x < minv ? minv : (x > maxv ? maxv : x);
  }
  for (uint32_t i = tid; i < NB + 8; i += kSelThr) P[i] = 0;
  for (uint32_t i = tid; i < NB / 32; i += kSelThr) need[i] = 0;
  if (tid < 4) big[kSelBigCap + tid] = 0;
  block_sort_sample<L>(skey, srt);
  SEL_STAMP(0);
  // ---- (B) segments: lower bounds at sample quantiles -- every 19th sample in the interior, and geometrically closer (8, 4, 2, 1
  //      samples from either end) in the tails, where a power law would otherwise pile a whole segment's population into its
  //      first bucket.  Two equal neighbouring quantiles are a heavy value, which gets a segment of its own, [v, v + 1) ----
  const uint32_t range_bl = bitlen<L>((L)(maxv - minv));
  const uint32_t cell_sh = range_bl > kSelCellLog ? range_bl - kSelCellLog : 0u;   // value cells: the top 11 bits of x - min
  if (tid == 0) {   // (par holds 1 for the bounds of a heavy value's segment until the scaling is written below)
    uint32_t ns = 0; L last_q = minv;
    seg[ns].lo = minv; seg[ns++].par = 0;
    auto cand = [&](uint32_t idx) {
      const L q = srt[idx];
      if (q == last_q) { if (seg[ns - 1].lo == q && q < maxv && ns < kSelSegs) { seg[ns - 1].par = 1; seg[ns].lo = (L)(q + 1); seg[ns++].par = 1; } }
      else if (q > seg[ns - 1].lo && ns < kSelSegs) { seg[ns].lo = q; seg[ns++].par = 0; }
      last_q = q;
    };
    cand(1); cand(2); cand(4); cand(8);
    for (uint32_t idx = 19; idx + 8 < kSelSample; idx += 19) cand(idx);
    cand(kSelSample - 8); cand(kSelSample - 4); cand(kSelSample - 2); cand(kSelSample - 1);
    big[kSelBigCap + 2] = ns;
  }
  __syncthreads();
  // ---- (B') data that is spread evenly over its range (incompressible columns: BASELINE configs[0], ids, hashes) needs no quantile map: 64-128
  //      EQUAL power-of-two segments put 30-60 latents into every bucket, and a latent's bucket is one shift -- no cell table, no search, no
  //      segment record: the count pass was bound by those random LDS reads (scripts/sel_timing.py: 377 k of 1.18 M cycles a variable).  Taken when
  //      the sorted sample says so: no heavy value, and no power-of-two segment holds more than three times its share of the sample.  The segment
  //      table is rewritten in the same form, so everything behind the count pass (and its tail loop) goes through the tables as before. ----
  const uint32_t flat_w = range_bl >= 15 ? range_bl - 7 : 8u;   // (ranges here are >= 32768: range_bl >= 16)
  const uint32_t flat_segs = (uint32_t)((uint64_t)(L)(maxv - minv) >> flat_w) + 1u;   // 64 .. 128
  {
    bool bad = false;
    if (tid < flat_segs) {
      const L lo = (L)(minv + ((L)tid << flat_w));
      L hi = (L)(lo + (L)(((L)1 << flat_w) - 1)); if (hi > maxv || hi < lo) hi = maxv;
      uint32_t a = 0, b = kSelSample;   // first sample >= lo
      while (a < b) { const uint32_t mid = (a + b) >> 1; if (srt[mid] < lo) a = mid + 1; else b = mid; }
      uint32_t c = a, e = kSelSample;   // first sample > hi
      while (c < e) { const uint32_t mid = (c + e) >> 1; if (srt[mid] <= hi) c = mid + 1; else e = mid; }
      bad = (c - a) * flat_segs > 3u * kSelSample;
    }
    if (tid < uni(big[kSelBigCap + 2]) && seg[tid].par == 1) bad = true;   // a heavy value
    // (a range just above a power of two makes 64 segments of twice the population: windows beyond the 64 latents a wave orders in registers -- the quantile map keeps those)
    const bool flat_now = __syncthreads_or(bad ? 1 : 0) == 0 && range_bl >= 16 && flat_segs >= 90;
    if (flat_now) {
      if (tid < flat_segs) { seg[tid].lo = (L)(minv + ((L)tid << flat_w)); seg[tid].par = 0; }
      if (tid == 0) { big[kSelBigCap + 2] = flat_segs; big[kSelBigCap + 3] = 1; }
    }
    __syncthreads();
  }
  const bool flat = uni(big[kSelBigCap + 3]) != 0;
  const uint32_t n_seg = uni(big[kSelBigCap + 2]);
  // A bound that is alone in its value cell moves down to the cell's first value (any monotone map will do): the cell then lies in ONE
  // segment and its latents need no search in the count pass.  On smooth data that is nearly every cell; where the data is dense
  // (several bounds in a cell) and around heavy values the bounds stay where the sample put them.
  {
    L lo = (L)0;
    if (tid < n_seg) {
      lo = seg[tid].lo;
      if (tid >= 1 && seg[tid].par == 0) {
        const uint32_t c = (uint32_t)((L)(lo - minv) >> cell_sh), pc = (uint32_t)((L)(seg[tid - 1].lo - minv) >> cell_sh);
        const uint32_t nc = tid + 1 < n_seg ? (uint32_t)((L)(seg[tid + 1].lo - minv) >> cell_sh) : 0xffffffffu;
        if (pc != c && nc != c) lo = (L)(minv + ((L)c << cell_sh));
      }
    }
    __syncthreads();
    if (tid < n_seg) seg[tid].lo = lo;
    __syncthreads();
    if (tid < n_seg) {
      const L last = tid + 1 < n_seg ? (L)(seg[tid + 1].lo - 1) : maxv;     // last value of the segment
      const L w1 = flat ? (L)(((L)1 << flat_w) - 1) : (L)(last - lo);         // width - 1 (flat: every segment scales as a full one, the last included)
      // sub-bucket = ((x - lo) >> pre) * m >> 16, monotone and < 64: segments of at most 64 values get one bucket per value
      // (pre 0, m 65536: "exact"), the others spread their (pre-shifted, < 2^16) width over all 64 sub-buckets
      uint32_t pre = 0, m = 65536;
      if ((uint64_t)w1 >= (1u << kSelSubLog)) {
        const uint32_t bl = bitlen<L>(w1);
        pre = bl > 16 ? bl - 16 : 0u;
        const uint32_t vmax = (uint32_t)(w1 >> pre);
        m = (uint32_t)(((uint64_t)1 << (16 + kSelSubLog)) / ((uint64_t)vmax + 1));
      }
      seg[tid].par = m | (pre << 24);
    }
  }
  __syncthreads();
  // value cells (the top 11 bits of x - min) -> the segments a cell meets: most cells meet one (the bounds were moved to cell starts
  // above), and the search below only walks the segments of the latent's cell.  (256 cells left skewed data -- a power law, the delta of
  // anything -- with most of its segments in one or two cells and up to seven search steps per latent.)
  for (uint32_t cell = tid; cell < kSelCells; cell += kSelThr) {
    auto seg_of = [&](L x) { uint32_t lo = 0, hi = n_seg; while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (seg[mid].lo <= x) lo = mid; else hi = mid; } return lo; };
    const L c0 = (L)(minv + ((L)cell << cell_sh));
    L c1 = (L)(c0 + (L)(((L)1 << cell_sh) - 1));
    const bool in = (uint64_t)cell <= ((uint64_t)(L)(maxv - minv) >> cell_sh);
    if (c1 > maxv || c1 < c0) c1 = maxv;
    lut[cell] = in ? (uint16_t)(seg_of(c0) | (seg_of(c1) << 8)) : (uint16_t)0;
  }
  __syncthreads();
  auto bucket_of = [&](L x) {   // monotone in x
    const uint32_t e = lut[(uint32_t)((L)(x - minv) >> cell_sh)];
    uint32_t j = e & 0xffu, jn = e >> 8;
    while (__any(j < jn)) {   // last segment of [j, jn] whose lower bound is <= x
      const uint32_t mid = (j + jn + 1) >> 1;
      const bool ge = seg[mid].lo <= x;
      j = ge ? mid : j; jn = ge ? jn : mid - 1;
    }
    const L rlo = seg[j].lo; const uint32_t par = seg[j].par;
    const uint32_t v = (uint32_t)((L)(x - rlo) >> (par >> 24));
    return (j << kSelSubLog) + ((v * (par & 0x1ffffu)) >> 16);
  };
  auto bucket_exact = [&](uint32_t k) { return seg[k >> kSelSubLog].par == 65536u; };
  auto bucket_value = [&](uint32_t k) { return (L)(seg[k >> kSelSubLog].lo + (L)(k & ((1u << kSelSubLog) - 1))); };   // of an exact bucket
  // ---- (C) count; the bucket of every latent is kept (u16) for the gather pass.  A thread owns kE consecutive latents per
  //      round, fetched with 16-byte loads (64 bytes in flight per thread: dword loads left the block at a third of what HBM gives
  //      one CU); every stage of the kE searches is issued together ----
  constexpr uint32_t kE = sizeof(L) == 8 ? 8u : 16u, kV = kE * sizeof(L) / 16;   // latents / 16-byte vectors per thread and round
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  union VecL { u32x4 v[kV]; L x[kE]; };
  auto count_pass = [&](auto flat_c) {
    constexpr bool kFlat = decltype(flat_c)::value;
    const uint32_t fsh = flat_w - kSelSubLog;
    uint32_t base = 0;
    VecL nxt;   // the next round's latents are fetched before this round's are worked on (the block's waves run in step: without it
                // they all wait for HBM together, then all compute together)
    if (kE * kSelThr <= n_all) {
#pragma unroll
      for (uint32_t q = 0; q < kV; q++) nxt.v[q] = ((const u32x4 PCO_GLOBAL*)(lat + tid * kE))[q];
    }
    for (; base + kE * kSelThr <= n_all; base += kE * kSelThr) {
      VecL d = nxt; uint32_t bb[kE];
      const uint32_t i0 = base + tid * kE;
      if (base + 2 * kE * kSelThr <= n_all) {
#pragma unroll
        for (uint32_t q = 0; q < kV; q++) nxt.v[q] = ((const u32x4 PCO_GLOBAL*)(lat + i0 + kE * kSelThr))[q];
      }
      if constexpr (kFlat) {   // segment = (x - min) >> w, sub-bucket = the next six bits: what the tables say, without reading them
        // (no clamp: a stored latent lies in [min, max], and what an unstored position makes of this is replaced below)
#pragma unroll
        for (uint32_t k = 0; k < kE; k++) bb[k] = (uint32_t)((L)(d.x[k] - minv) >> fsh);
      } else {
#pragma unroll
        for (uint32_t k = 0; k < kE; k++) d.x[k] = d.x[k] < minv ? minv : (d.x[k] > maxv ? maxv : d.x[k]);   // (positions that are not stored may hold anything, and the tables are indexed with it)
        uint32_t j[kE], jn[kE];
#pragma unroll
        for (uint32_t k = 0; k < kE; k++) {
          const uint32_t e = lut[(uint32_t)((L)(d.x[k] - minv) >> cell_sh)];
          j[k] = e & 0xffu; jn[k] = e >> 8;
        }
        for (;;) {
          bool more = false;
#pragma unroll
          for (uint32_t k = 0; k < kE; k++) more = more || j[k] < jn[k];
          if (!__any(more)) break;
#pragma unroll
          for (uint32_t k = 0; k < kE; k++) {
            const uint32_t mid = (j[k] + jn[k] + 1) >> 1;
            const bool ge = seg[mid].lo <= d.x[k];
            j[k] = ge ? mid : j[k]; jn[k] = ge ? jn[k] : mid - 1;
          }
        }
        // (staged: all segment records, then all buckets, then the stored-or-not flags, then the counter bumps -- one loop over the
        //  latents made every LDS round trip wait for the one before it)
        L rlo[kE]; uint32_t par[kE];
#pragma unroll
        for (uint32_t k = 0; k < kE; k++) { rlo[k] = seg[j[k]].lo; par[k] = seg[j[k]].par; }
#pragma unroll
        for (uint32_t k = 0; k < kE; k++) {
          const uint32_t v = (uint32_t)((L)(d.x[k] - rlo[k]) >> (par[k] >> 24));
          bb[k] = (j[k] << kSelSubLog) + ((v * (par[k] & 0x1ffffu)) >> 16);
        }
      }
      if (skip != 0) {   // (block-uniform)
#pragma unroll
        for (uint32_t k = 0; k < kE; k++) bb[k] = stored(i0 + k) ? bb[k] : 0xffffu;
      }
#pragma unroll
      for (uint32_t k = 0; k < kE; k++) if (bb[k] != 0xffffu) atomicAdd((uint32_t*)&P[bb[k]], 1u);
      if constexpr (!kFlat) {   // (a flat map's gather pass shifts again instead of reading the id back: half a MB less each way per variable)
#pragma unroll
        for (uint32_t q = 0; q < kE / 8; q++) {   // eight u16 ids per 16-byte store
          u32x4 w; w.x = bb[8 * q] | (bb[8 * q + 1] << 16); w.y = bb[8 * q + 2] | (bb[8 * q + 3] << 16); w.z = bb[8 * q + 4] | (bb[8 * q + 5] << 16); w.w = bb[8 * q + 6] | (bb[8 * q + 7] << 16);
          ((u32x4 PCO_GLOBAL*)(ids + i0))[q] = w;
        }
      }
    }
    for (uint32_t i0 = base; i0 < n_all; i0 += kSelThr) {   // (whole waves run bucket_of: its search loop votes)
      const uint32_t i = i0 + tid;
      const bool on = i < n_all && stored(i);
      L xv = i < n_all ? lat[i] : minv; xv = xv < minv ? minv : (xv > maxv ? maxv : xv);
      const uint32_t b = bucket_of(xv);
      if (!kFlat && i < n_all) ids[i] = on ? (uint16_t)b : (uint16_t)0xffffu;
      if (on) atomicAdd((uint32_t*)&P[b], 1u);
    }
  };
  if (flat) count_pass(BoolC<true>{}); else count_pass(BoolC<false>{});
  __threadfence_block();
  __syncthreads();
  SEL_STAMP(1);
  // ---- (D) exclusive prefix over the buckets ----
  {
    constexpr uint32_t PER = NB / kSelThr;
    uint32_t s0 = 0;
    for (uint32_t k = 0; k < PER; k++) s0 += P[tid * PER + k];
    const uint32_t incl = wave_incl_scan(s0);
    if (lane == 63) scan[wave] = incl;
    __syncthreads();
    uint32_t wbase = 0; for (uint32_t w = 0; w < wave; w++) wbase += scan[w];
    uint32_t run = wbase + incl - s0;
    for (uint32_t k = 0; k < PER; k++) { const uint32_t c = P[tid * PER + k]; P[tid * PER + k] = run; run += c; }
    if (tid == kSelThr - 1) P[NB] = run;   // == n_lat
  }
  __syncthreads();
  auto bucket_of_rank = [&](uint32_t r) {   // the (non-empty) bucket holding rank r: last k with P[k] <= r
    uint32_t lo = 0, hi = NB;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (P[mid] <= r) lo = mid; else hi = mid; }
    return lo;
  };
  // ---- (E) mark the buckets that hold a queried rank, and the nearest non-empty bucket on either side; exact buckets answer
  //      from their counts and are not gathered ----
  auto mark = [&](uint32_t k) { if (!bucket_exact(k)) atomicOr((uint32_t*)&need[k >> 5], 1u << (k & 31)); };
  if (tid < B) {
    const uint32_t c = c_count(tid);
    for (uint32_t which = 0; which < 2; which++) {
      const uint32_t r = c - 1 + which;
      if (r >= n_lat) continue;
      const uint32_t k = bucket_of_rank(r);
      mark(k);
      if (P[k] > 0) mark(bucket_of_rank(P[k] - 1));
      if (P[k + 1] < n_lat) mark(bucket_of_rank(P[k + 1]));
    }
  }
  __syncthreads();
  // ---- (F) list of marked buckets (ascending) with the first subset index of each one's window ----
  uint32_t n_need = 0, n_sub = 0;
  {
    uint32_t word = 0, nb_here = 0, el_here = 0;
    if (tid < NB / 32) {
      word = need[tid]; nb_here = __popc(word);
      for (uint32_t w = word; w; w &= w - 1) { const uint32_t k = tid * 32 + (uint32_t)__builtin_ctz(w); el_here += P[k + 1] - P[k]; }
    }
    const uint32_t i_nb = wave_incl_scan(nb_here), i_el = wave_incl_scan(el_here);
    if (lane == 63) { scan[wave] = i_nb; scan[16 + wave] = i_el; }
    __syncthreads();
    uint32_t b_nb = 0, b_el = 0; for (uint32_t w = 0; w < wave; w++) { b_nb += scan[w]; b_el += scan[16 + w]; }
    uint32_t slot = b_nb + i_nb - nb_here, off = b_el + i_el - el_here;
    for (uint32_t w = word; w; w &= w - 1) {
      const uint32_t k = tid * 32 + (uint32_t)__builtin_ctz(w);
      if (slot < kSelMaxNeeded) nl_oc[slot + 1] = off;   // window `slot` starts at off: entry slot + 1 is its fill cursor in (G), which leaves it at the window's end = the next one's start
      slot++; off += P[k + 1] - P[k];
    }
    if (tid < NB / 32) wpre[tid] = (uint16_t)(b_nb + i_nb - nb_here);
    if (tid == 0) nl_oc[0] = 0;
    n_need = scan[0] + scan[1] + scan[2] + scan[3]; n_sub = scan[16] + scan[17] + scan[18] + scan[19];   // (the bitmap's 256 words live in waves 0..3)
  }
  __syncthreads();
  if (n_need > kSelMaxNeeded) {   // cannot happen (<= 6 marks per histogram bin); the radix-sort kernel would deal with it
    if (tid == 0) ev->hist_path = 2;
    __syncthreads();
    return;
  }
  L PCO_GLOBAL* S = sort_ptr<L>(ws, t, 1);
  SEL_STAMP(2);
  // ---- (G) gather: every latent of a marked bucket goes into its bucket's window, in any order.  Same thread-to-latent mapping
  //      and load width as the count pass; staged (window lookups, then cursor atomics, then stores) so that the LDS round trips
  //      of a thread's latents overlap ----
  // (kMasked: some id may be 0xffff, "not stored" -- always with the ids of the quantile map, with a flat map only when the variable has unstored
  //  positions.  The gather pass is bound by its VALU instructions, ~30 per latent at sixteen waves a CU: the flat map's do without the clamp and,
  //  unmasked, without the test)
  auto gather_pass = [&](auto flat_c, auto masked_c) {
    constexpr bool kFlat = decltype(flat_c)::value, kMasked = decltype(masked_c)::value;
    const uint32_t fsh = flat_w - kSelSubLog;
    auto flat_id = [&](L x, bool on) { return on ? (uint32_t)((L)(x - minv) >> fsh) : 0xffffu; };   // (as the count pass counted it: a stored latent lies in [min, max])
    uint32_t base = 0;
    union IdV { u32x4 v[kE / 8]; uint16_t k[kE]; };
    VecL nxt; IdV nid;
    if (kE * kSelThr <= n_all) {
#pragma unroll
      for (uint32_t q = 0; q < kV; q++) nxt.v[q] = ((const u32x4 PCO_GLOBAL*)(lat + tid * kE))[q];
      if constexpr (!kFlat) {
#pragma unroll
        for (uint32_t q = 0; q < kE / 8; q++) nid.v[q] = ((const u32x4 PCO_GLOBAL*)(ids + tid * kE))[q];
      }
    }
    for (; base + kE * kSelThr <= n_all; base += kE * kSelThr) {
      VecL d = nxt; IdV id = nid; uint32_t sl[kE], at[kE], kid[kE];
      const uint32_t i0 = base + tid * kE;
      if (base + 2 * kE * kSelThr <= n_all) {
#pragma unroll
        for (uint32_t q = 0; q < kV; q++) nxt.v[q] = ((const u32x4 PCO_GLOBAL*)(lat + i0 + kE * kSelThr))[q];
        if constexpr (!kFlat) {
#pragma unroll
          for (uint32_t q = 0; q < kE / 8; q++) nid.v[q] = ((const u32x4 PCO_GLOBAL*)(ids + i0 + kE * kSelThr))[q];
        }
      }
      if constexpr (kFlat) {
#pragma unroll
        for (uint32_t k = 0; k < kE; k++) kid[k] = flat_id(d.x[k], true);
        if (skip != 0) {   // (block-uniform)
#pragma unroll
          for (uint32_t k = 0; k < kE; k++) kid[k] = stored(i0 + k) ? kid[k] : 0xffffu;
        }
      } else {
#pragma unroll
        for (uint32_t k = 0; k < kE; k++) kid[k] = id.k[k];
      }
#pragma unroll
      for (uint32_t k = 0; k < kE; k++) at[k] = (kid[k] & (kSelBuckets - 1)) >> 5;   // bitmap word (an id of 0xffff -- not stored -- reads the last one)
      uint32_t wd[kE];
#pragma unroll
      for (uint32_t k = 0; k < kE; k++) { wd[k] = need[at[k]]; sl[k] = wpre[at[k]]; }   // unconditional reads, all in flight together
#pragma unroll
      for (uint32_t k = 0; k < kE; k++) {
        const uint32_t kk = kid[k];
        const uint32_t hit = (kMasked ? (uint32_t)(kk != 0xffffu) : 1u) & (wd[k] >> (kk & 31)) & 1u;   // (bitwise: no short-circuit branch per latent)
        const uint32_t slot = sl[k] + (uint32_t)__popc(wd[k] & ((1u << (kk & 31)) - 1u));
        sl[k] = hit ? slot : 0xffffu;
      }
#pragma unroll
      for (uint32_t k = 0; k < kE; k++) at[k] = sl[k] != 0xffffu ? atomicAdd((uint32_t*)&nl_oc[sl[k] + 1], 1u) : 0u;
#pragma unroll
      for (uint32_t k = 0; k < kE; k++) if (sl[k] != 0xffffu) S[at[k]] = d.x[k];
    }
    for (uint32_t i = base + tid; i < n_all; i += kSelThr) {
      const uint32_t k = kFlat ? flat_id(lat[i], stored(i)) : (uint32_t)ids[i];
      if (k != 0xffffu && ((need[k >> 5] >> (k & 31)) & 1u)) S[atomicAdd((uint32_t*)&nl_oc[slot_of(k) + 1], 1u)] = lat[i];
    }
  };
  if (!flat) gather_pass(BoolC<false>{}, BoolC<true>{}); else if (skip != 0) gather_pass(BoolC<true>{}, BoolC<true>{}); else gather_pass(BoolC<true>{}, BoolC<false>{});
  __threadfence_block();
  __syncthreads();
  SEL_STAMP(3);
  // ---- (H) order every window: one value -> nothing to do; up to 64 latents -> one wave, in registers; up to 8192 -> the
  //      block, in LDS; more -> the fallback kernel ----
  {
    L PCO_LDS* wsrt = srt + wave * kWaveSortCap;   // this wave's private sort area
    constexpr uint32_t kWaves = kSelThr / 64, kGrp = 4;
    uint32_t noc[kGrp], nlen[kGrp]; L nfirst[kGrp];
    auto fetch = [&](uint32_t s0) {   // the first 64 latents of four windows
#pragma unroll
      for (uint32_t g = 0; g < kGrp; g++) {
        const uint32_t s = s0 + g;
        noc[g] = s < n_need ? nl_oc[s] : 0u; nlen[g] = s < n_need ? nl_oc[s + 1] - noc[g] : 0u;
        nfirst[g] = lane < nlen[g] ? S[noc[g] + lane] : (L)0;
      }
    };
    fetch(wave * kGrp);
    for (uint32_t s0 = wave * kGrp; s0 < n_need; s0 += kWaves * kGrp) {   // four windows per round, the next round's already in flight
      uint32_t oc[kGrp], len[kGrp]; L first[kGrp];
#pragma unroll
      for (uint32_t g = 0; g < kGrp; g++) { oc[g] = noc[g]; len[g] = nlen[g]; first[g] = nfirst[g]; }
      fetch(s0 + kWaves * kGrp);
      // windows of up to 64 latents (nearly all of them): padded with the window's maximum, the four sorted in lockstep
      L key[kGrp]; bool small_multi[kGrp];
#pragma unroll
      for (uint32_t g = 0; g < kGrp; g++) {
        small_multi[g] = len[g] > 1 && len[g] <= 64;   // (wave-uniform)
        key[g] = lane < len[g] ? first[g] : (L)~(L)0;
      }
      if (small_multi[0] || small_multi[1] || small_multi[2] || small_multi[3]) {
        wave_sort64_multi<L, kGrp>(key);
#pragma unroll
        for (uint32_t g = 0; g < kGrp; g++) if (small_multi[g] && lane < len[g]) S[oc[g] + lane] = key[g];
      }
#pragma unroll
      for (uint32_t g = 0; g < kGrp; g++) {
        const uint32_t n_w = len[g];
        if (n_w <= 64) continue;
        L mn = first[g], mx = first[g];
        if (n_w <= kWaveSortCap) wsrt[lane] = first[g];
        for (uint32_t i = 64 + lane; i < n_w; i += 64) { const L x = S[oc[g] + i]; mn = x < mn ? x : mn; mx = x > mx ? x : mx; if (n_w <= kWaveSortCap) wsrt[i] = x; }
        mn = wave_butterfly(mn, [](L p, L q) { return p < q ? p : q; }); mx = wave_butterfly(mx, [](L p, L q) { return p > q ? p : q; });
        if (mn == mx) continue;   // one value: in order as it is
        if (n_w <= kWaveSortCap) {
          uint32_t p2 = 128; while (p2 < n_w) p2 <<= 1;
          for (uint32_t i = n_w + lane; i < p2; i += 64) wsrt[i] = mx;
          enc_wave_sync();
          wave_sort_lds<L>(wsrt, p2);
          for (uint32_t i = lane; i < n_w; i += 64) S[oc[g] + i] = wsrt[i];
          enc_wave_sync();
        } else if (lane == 0) {
          if (n_w > kBlockSortCap) big[kSelBigCap + 1] = 1;
          else { const uint32_t at = atomicAdd((uint32_t*)&big[kSelBigCap], 1u); if (at < kSelBigCap) big[at] = s0 + g; else big[kSelBigCap + 1] = 1; }
        }
      }
    }
  }
  __threadfence_block();
  __syncthreads();
  if (big[kSelBigCap + 1] != 0) {
    if (tid == 0) ev->hist_path = 2;
    __syncthreads();
    return;
  }
  SEL_STAMP(4);
  const uint32_t n_big = big[kSelBigCap];
  for (uint32_t bi = 0; bi < n_big; bi++) {
    const uint32_t s = big[bi], oc = nl_oc[s], len = nl_oc[s + 1] - oc;
    uint32_t p2 = 128; while (p2 < len) p2 <<= 1;
    for (uint32_t i = tid; i < p2; i += kSelThr) srt[i] = i < len ? S[oc + i] : maxv;
    __syncthreads();
    block_sort_lds<L>(srt, p2);
    for (uint32_t i = tid; i < len; i += kSelThr) S[oc + i] = srt[i];
    __syncthreads();
  }
  __threadfence_block();
  __syncthreads();
  SEL_STAMP(5);
  // ---- (I) rank queries (as in the radix-sort path: runs of equal values never leave their bucket) ----
  auto value_at = [&](uint32_t r) {
    const uint32_t k = bucket_of_rank(r);
    if (bucket_exact(k)) return bucket_value(k);
    return S[nl_oc[slot_of(k)] + (r - P[k])];
  };
  auto lookup = [&](uint32_t r, L& value, uint32_t& st, uint32_t& en) {
    const uint32_t k = bucket_of_rank(r);
    if (bucket_exact(k)) { value = bucket_value(k); st = P[k]; en = P[k + 1]; return; }
    const uint32_t oc = nl_oc[slot_of(k)];
    const uint32_t at = oc + (r - P[k]), wend = oc + (P[k + 1] - P[k]);
    value = S[at];
    uint32_t lo = oc, hi = at;   // first index with S[idx] >= value
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (S[mid] < value) lo = mid + 1; else hi = mid; }
    st = P[k] + (lo - oc);
    lo = at + 1; hi = wend;      // first index with S[idx] > value
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (S[mid] <= value) lo = mid + 1; else hi = mid; }
    en = P[k] + (lo - oc);
  };
  if (tid < B) {
    const uint32_t c = c_count(tid);
    L v; uint32_t st, en; lookup(c - 1, v, st, en);
    rv[tid] = v; rst[tid] = st; ren[tid] = en;
    rnext[tid] = c < n_lat ? value_at(c) : (L)0;
    rpred[tid] = st > 0 ? value_at(st - 1) : (L)0;
    rsucc[tid] = en < n_lat ? value_at(en) : (L)0;
  }
  __syncthreads();
  SEL_STAMP(6);
  hist_emit<L>(n_lat, bins_log, minv, rv, rst, ren, rnext, rpred, rsucc, plan, ev, 1u, (uint32_t PCO_LDS*)(smem + kSelLdsP),
               ws.walk != nullptr ? (uint8_t PCO_GLOBAL*)ws.walk + ((uint64_t)t * 3 + var) * kWalkRecBytes : (uint8_t PCO_GLOBAL*)nullptr);
  __syncthreads();
  SEL_STAMP(7);
#ifdef PCO_SEL_TIMING
  if (tid == 0) { atomicAdd(&g_sel_timing[8], 1ull); atomicAdd(&g_sel_timing[9], (unsigned long long)n_need); atomicAdd(&g_sel_timing[10], (unsigned long long)n_sub); atomicAdd(&g_sel_timing[11], (unsigned long long)n_big); atomicAdd(&g_sel_timing[12], (unsigned long long)n_seg); }
#endif
}

// grid = chunks, 512 threads: every wide-range variable of the chunks whose latents are L (one instantiation per latent width, launched
// for the widths the call holds: as one kernel over all four the variable bodies were out-of-line calls with the workspace on the stack
// and their callee-saved registers spilled around the streaming loops).  The delta latent variable (lookbacks < 2^15) never has a wide range.
template <class L>
__global__ __launch_bounds__(kSelThr, 4) void enc_hist_select_kernel(EncWorkspace ws, uint32_t n_tasks) {
  const uint32_t t = blockIdx.x;
  if (t >= n_tasks) return;
  const EncChunk PCO_GLOBAL* ch = (const EncChunk PCO_GLOBAL*)ws.chunks + t;
  if (uni(ch->status) != PCO_GFX_OK || uni(ch->big) != 0) return;   // (more than 256 bins: the sort kernel)
  if ((uint32_t)dtype_bits(uni(ch->dtype)) != LBits<L>::v) return;
  const uint32_t ubl = uni(ch->unopt_bins_log);
  for (uint32_t var = 1; var < 3; var++) {
    if (!uni(ch->v[var].present)) continue;
    select_var<L>(ws, t, var, var == 2 ? (ubl < 6 ? ubl : 6) : ubl);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// enc_hist_small_kernel: variables of at most kSmallHistCap latents whose value range is beyond enc_hist_kernel's counters (>= 4096) (the 6.6 k-latent samples of the Auto-delta trials,
// short chunks).  The whole variable is ordered in LDS: keys are x - min, 32 bits wide whenever the range allows, sorted in
// place by block_radix_sort_inplace, and the <= 256 rank queries read the sorted array directly.  LDS = the record area +
// n keys (launcher: small_lds_bytes), so two to four blocks share a CU and hide each other's barriers and loads.
// ---------------------------------------------------------------------------------------------------------------------------
__host__ __device__ constexpr uint32_t small_lds_bytes(uint32_t n, uint32_t key_bytes) { return kHistLdsCounts + (n * key_bytes > kWalkScratchBytes + 16u ? ((n * key_bytes + 15u) & ~15u) : kWalkScratchBytes + 16u); }   // (at least hist_emit's scratch)

template <class L, class K>
__device__ __forceinline__ void small_body(const EncWorkspace& ws, uint32_t t, uint32_t var, uint32_t bins_log, EncChunk PCO_GLOBAL* ch, EncVar PCO_GLOBAL* ev,
                                           uint32_t n_lat, L minv, L maxv) {
  const PlanRef plan = plan_ref(ws, t, var);
  const uint32_t tid = threadIdx.x, lane = lane_id();
  const L PCO_GLOBAL* lat = lat_ptr<L>(ws, t, var);
  const uint32_t n_all = (uint32_t)ch->n, skip = ev->lat_start, plow = ch->page_low, pr = ch->page_r;
  const bool single_page = ch->n_pages == 1;
  const bool exact_paging = ch->exact_paging != 0; const uint32_t n_pg = ch->n_pages;
  const EncPage PCO_GLOBAL* pgl = (const EncPage PCO_GLOBAL*)ws.pages + ch->page_first;
  auto exact_start = [&](uint32_t i) {
    uint32_t lo = 0, hi = n_pg;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if ((uint64_t)pgl[mid].start <= i) lo = mid; else hi = mid; }
    return (uint64_t)pgl[lo].start;
  };
  auto stored = [&](uint32_t i) { return skip == 0 || (single_page ? i >= skip : (uint64_t)i - (exact_paging ? exact_start(i) : page_start_of(i, plow, pr)) >= skip); };
  uint8_t PCO_LDS* smem = enc_lds_base();
  L PCO_LDS* rv = (L PCO_LDS*)(smem + kHistLdsRecV);
  L PCO_LDS* rnext = (L PCO_LDS*)(smem + kHistLdsRecV + 2048);
  L PCO_LDS* rpred = (L PCO_LDS*)(smem + kHistLdsRecV + 4096);
  L PCO_LDS* rsucc = (L PCO_LDS*)(smem + kHistLdsRecV + 6144);
  uint32_t PCO_LDS* rst = (uint32_t PCO_LDS*)(smem + kHistLdsRecV + 8192);
  uint32_t PCO_LDS* ren = (uint32_t PCO_LDS*)(smem + kHistLdsRecV + 9216);
  uint32_t PCO_LDS* cursor = (uint32_t PCO_LDS*)(smem + kRadixScratchBytes);
  K PCO_LDS* srt = (K PCO_LDS*)(smem + kHistLdsCounts);
  const uint32_t B = 1u << bins_log;
  const uint64_t n64 = n_lat;
  auto c_count = [&](uint32_t b) { return (uint32_t)((((uint64_t)b + 1) * n64 + B - 1) >> bins_log); };
  __syncthreads();   // (the previous variable's records are done with)
  // A chunk whose split left 16-bit latents (EncChunk::c16_ok == 1; its ranges are below 32768) has no full-width latents: like the
  // counting kernels, read the 16-bit ones (relative to c16_ref) and report the compact layout (hist_path 0).
  const bool c16 = uni(ch->c16_ok) == 1 && var != 0;
  uint16_t PCO_GLOBAL* clat = clat_ptr(ws, t, var);
  const uint16_t c16_off = (uint16_t)((uint64_t)minv - (uint64_t)ch->c16_ref[var == 2 ? 1 : 0]);
  auto value_at = [&](uint32_t i) { return c16 ? (K)(uint16_t)(clat[i] - c16_off) : (K)(L)(lat[i] - minv); };
  if (single_page) {   // stored latents = positions skip ..
    for (uint32_t i = skip + tid; i < n_all; i += kSelT) srt[i - skip] = value_at(i);
  } else {             // any order will do: one cursor bump per wave and round
    if (tid == 0) *cursor = 0;
    __syncthreads();
    for (uint32_t i0 = 0; i0 < n_all; i0 += kSelT) {
      const uint32_t i = i0 + tid;
      const bool on = i < n_all && stored(i);
      const uint64_t m = __ballot(on);
      uint32_t at = 0;
      if (lane == 0 && m) at = atomicAdd((uint32_t*)cursor, (uint32_t)__popcll(m));
      at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
      if (on) srt[at + __popcll(m & (((uint64_t)1 << lane) - 1))] = value_at(i);
    }
  }
  __syncthreads();
  block_radix_sort_inplace<K>(srt, smem + kHistLdsRecV, n_lat, bitlen<L>((L)(maxv - minv)));
  if (tid < B) {
    const uint32_t c = c_count(tid);
    const K v = srt[c - 1];
    uint32_t lo = 0, hi = c - 1;   // first index with srt[idx] >= v
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (srt[mid] < v) lo = mid + 1; else hi = mid; }
    const uint32_t st = lo;
    lo = c; hi = n_lat;            // first index with srt[idx] > v
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (srt[mid] <= v) lo = mid + 1; else hi = mid; }
    const uint32_t en = lo;
    const K nx = c < n_lat ? srt[c] : (K)0, pd = st > 0 ? srt[st - 1] : (K)0, sc = en < n_lat ? srt[en] : (K)0;
    rv[tid] = (L)(minv + (L)v); rst[tid] = st; ren[tid] = en;
    rnext[tid] = c < n_lat ? (L)(minv + (L)nx) : (L)0;
    rpred[tid] = st > 0 ? (L)(minv + (L)pd) : (L)0;
    rsucc[tid] = en < n_lat ? (L)(minv + (L)sc) : (L)0;
  }
  __syncthreads();
  hist_emit<L>(n_lat, bins_log, minv, rv, rst, ren, rnext, rpred, rsucc, plan, ev, c16 ? 0u : 1u, (uint32_t PCO_LDS*)(smem + kHistLdsCounts),
               ws.walk != nullptr ? (uint8_t PCO_GLOBAL*)ws.walk + ((uint64_t)t * 3 + var) * kWalkRecBytes : (uint8_t PCO_GLOBAL*)nullptr);
}

template <class L>
__device__ void small_var(const EncWorkspace& ws, uint32_t t, uint32_t var, uint32_t bins_log) {
  EncChunk PCO_GLOBAL* ch = (EncChunk PCO_GLOBAL*)ws.chunks + t;
  EncVar PCO_GLOBAL* ev = &ch->v[var];
  const uint32_t n_lat = ev->n_lat;
  if (n_lat == 0 || n_lat > kSmallHistCap) return;
  const L minv = (L)ev->minv, maxv = (L)ev->maxv;
  const uint64_t range = (uint64_t)(L)(maxv - minv);
  if (range < kDirectHistRange) return;   // enc_hist_kernel's value-space counting owns it
  if constexpr (sizeof(L) == 8) { if (range >> 32) { small_body<L, uint64_t>(ws, t, var, bins_log, ch, ev, n_lat, minv, maxv); return; } }
  small_body<L, uint32_t>(ws, t, var, bins_log, ch, ev, n_lat, minv, maxv);
}

// grid = chunks, 1024 threads, dynamic LDS = small_lds_bytes(largest small variable, its key width)
__global__ __launch_bounds__(kSelT, 8) void enc_hist_small_kernel(EncWorkspace ws, uint32_t n_tasks) {
  const uint32_t t = blockIdx.x;
  if (t >= n_tasks) return;
  const EncChunk PCO_GLOBAL* ch = (const EncChunk PCO_GLOBAL*)ws.chunks + t;
  if (uni(ch->status) != PCO_GFX_OK || uni(ch->big) != 0) return;   // (more than 256 bins: the sort kernel)
  const int bits = dtype_bits(uni(ch->dtype));
  const uint32_t ubl = uni(ch->unopt_bins_log);
  for (uint32_t var = 0; var < 3; var++) {
    if (!uni(ch->v[var].present)) continue;
    const uint32_t bl = var == 2 ? (ubl < 6 ? ubl : 6) : ubl;
    if (var == 0) small_var<uint32_t>(ws, t, var, bl);
    else if (bits == 64) small_var<uint64_t>(ws, t, var, bl);
    else if (bits == 32) small_var<uint32_t>(ws, t, var, bl);
    else if (bits == 16) small_var<uint16_t>(ws, t, var, bl);
    else small_var<uint8_t>(ws, t, var, bl);
  }
}

#ifdef PCO_RADIX_PROBE
__global__ __launch_bounds__(kSelT, 8) void radix_probe32(uint32_t n, uint32_t sig) { block_radix_sort_inplace<uint32_t>((uint32_t PCO_LDS*)(enc_lds_base() + kHistLdsCounts), enc_lds_base(), n, sig); }
__global__ __launch_bounds__(kSelT, 8) void radix_probe64(uint32_t n, uint32_t sig) { block_radix_sort_inplace<uint64_t>((uint64_t PCO_LDS*)(enc_lds_base() + kHistLdsCounts), enc_lds_base(), n, sig); }
#endif
// A/B switch (PCO_GFX_NO_HIST_SELECT): hand every long wide-range variable to the radix-sort kernel
__global__ void enc_hist_flag_kernel(EncWorkspace ws, uint32_t n_tasks) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_tasks) return;
  EncChunk* ch = ws.chunks + t;
  if (ch->status != PCO_GFX_OK) return;
  for (uint32_t var = 0; var < 3; var++) if (ch->v[var].present && ch->v[var].n_lat > kSmallHistCap && ch->v[var].maxv - ch->v[var].minv >= kWideHistRange) ch->v[var].hist_path = 2;
}

}  // namespace pcogfx
