// encode_lookback.hip -- the lookback search (delta/lookback.rs:22-185) as a five-wave pipeline per page.
//
// choose_lookbacks is greedy and strictly element-ordered, but only a small part of an element's work depends on what its
// predecessors chose.  enc_lookback_kernel (encode_kernels.hip) gives a page to ONE wave, which walks a tile of 64 elements through
// hash proposals -> candidate latents -> decision rounds -> delta application: a ~12 k-cycle chain of dependent LDS / HBM round trips
// per tile with nothing else on the SIMD to hide it.  Here a page belongs to a workgroup of five waves, each owning one stage and all
// advancing one tile per step (s_barrier between steps), so a step costs the slowest stage, not their sum:
//
//   wave 0 / 1  H0 / H1  tile s     the three hash proposals from the fine (l) / coarse (l >> 8) last-index table (lookback.rs:22-64):
//                                    independent of every decision, ordered only against their own table's earlier updates
//   wave 2      C        tile s-1   the latent and leading-zero count of the twelve decision-independent candidates (6 brute force, 6 hashed)
//   wave 3      D        tile s-2   the decisions (find_best_lookback + the repeating slots + the counts, lookback.rs:67-159), a tile at
//                                    a time under the guess that every element repeats its predecessor's lookback (exact: see
//                                    lookback_page in encode_kernels.hip, whose formulation this stage keeps)
//   wave 4      A        tile s-3   lookback.rs:166-185: l[i] - l[i - lookback] + MID, the chosen lookbacks, the variables' ranges
//
// Hand-over through LDS: the latents of the last kRing positions (ring), the proposals (u16), the leading-zero counts (u8) and the
// chosen lookbacks, double / triple buffered by tile parity.
//
// The last-index tables are what the search's random accesses go to (six reads and two updates per element into 2 x 2^(w+1) entries),
// and with u32 entries a 2^18-number page owns 640 KB of them.  So:
//   * entries are u16, the position mod 2^16.  Only entries within window_n <= 2^15 positions matter (lookback.rs:52-56), and a sweep
//     every 2^14 positions rewrites every entry older than the window to "window_n + 1 positions old", so no entry ever ages past 2^16
//     and (position - entry) mod 2^16 IS its age.  Tables: 256 KB per page instead of 512;
//   * a launch keeps two pages per CU in flight (the grid is a pool of page slots, each block takes pages until none are left): beyond
//     that nothing is gained -- measured, scripts/lb_scaling.py: 256 pages 9.9 ms, 512 pages 17-18 ms, 1024 pages 34-39 ms whatever the
//     entry width; a CU completes one tile's stage work (~17 k busy wave-cycles) per ~5 k cycles however many pages share it;
//   * pages of at most 8192 numbers (the Auto-delta trial samples: thousands per call) need no sweep -- a position fits 13 bits.  They
//     only come here on request (the host's PCO_GFX_LB_PIPE_SMALL): five per CU are no faster than sixteen one-wave pages.
// Data on which the "repeats its predecessor" guess fails for nearly every element (small random integers: ~49 rounds per tile) makes stage D
// the whole cost; there one wave per page and sixteen pages per CU (enc_lookback_kernel) is the better shape, so a page that averages more
// than kLbAbortRounds rounds for a tile right after its opening is handed back to that kernel (redo list), which runs after this one.
#pragma once
// (included by pco_gfx.hip after encode_kernels.hip, whose workspace types and lookback helpers it uses)

namespace pcogfx {

// kProps: the hash proposals come from enc_lookback_hash_kernel (below) as six u16 streams per page; the two H waves are replaced by one
// loader wave (latents into the ring, proposals into the queue) and the kernel has no random global access left but far candidates' latents.
template <bool kSmall, bool kPropsT = false, bool kFastDT = false> struct LbPipe {
  static constexpr bool kProps = kPropsT;
  static constexpr bool kFastD = kFastDT;   // stage D takes the best of the brute-force and of the hashed proposals ready-made from the stages in front of it (below)
  static constexpr uint32_t kFront = kPropsT ? 1u : 2u;                 // waves in front of stage C: H0 / H1, or the loader
  static constexpr uint32_t kWaves = kFront + 3, kThreads = 64 * kWaves;
  static constexpr uint32_t kRing = kSmall ? 1024u : 2048u;            // latents of the last kRing positions (u64 each)
  static constexpr uint32_t kNear = kRing - 64 * kWaves;               // lookbacks below this are served from the ring by every stage (they run up to four tiles apart)
  static constexpr uint32_t kCounts = kSmall ? 8192u : 4096u;          // lookback_counts kept in LDS (small pages: all of them, as u16)
  typedef std::conditional_t<kSmall, uint16_t, uint32_t> CountT;
  static constexpr uint32_t kOffCounts = 0;
  static constexpr uint32_t kOffRing = kOffCounts + kCounts * sizeof(CountT);
  static constexpr uint32_t kOffPlb = kOffRing + kRing * 8;             // u16[3][6][64]
  static constexpr uint32_t kOffLz = kOffPlb + 3 * 6 * 64 * 2;          // u8[2][12][64]
  static constexpr uint32_t kOffLb = kOffLz + 2 * 12 * 64;              // u32[2][64], then the abort flag
  static constexpr uint32_t kOffGrp = kOffLb + 2 * 64 * 4 + 16;         // u32[2][2][64]: goodness | lookback << 8 of the best brute-force / hashed proposal (kFastD)
  static constexpr uint32_t kLdsBytes = kOffGrp + (kFastDT ? 2 * 2 * 64 * 4 : 0);   // 37-38 KB (four pages per CU) / 28.5 KB (five)
};
constexpr uint32_t kLbPipeSmallMaxPage = 8192;
constexpr uint32_t kLbSweepPeriod = 1u << 14;      // positions between two sweeps of the u16 tables (window_n + 1 + period + a tile < 2^16)
// The page's first kLbSeqTiles tiles are decided element by element, sixteen lanes = the sixteen proposals: nothing has a history there
// and every element decides differently whatever the data, so speculation only costs (all lookbacks are below kNear there: latents from
// the ring).  A tile among the next kLbAbortWindow that needs more than kLbAbortRounds rounds sends the page to enc_lookback_kernel.
constexpr uint32_t kLbSeqTiles = 8, kLbAbortWindow = 8, kLbAbortRounds = 24;

#ifdef PCO_LBP_TIMING
__device__ unsigned long long g_lbp_timing[16];
#endif

template <class L, class Cfg>
__device__ bool lookback_page_pipe(const EncWorkspace& ws, uint32_t t, EncPage PCO_GLOBAL* pg, uint16_t PCO_GLOBAL* hash_tbl, uint32_t PCO_GLOBAL* gcounts,
                                   const uint16_t PCO_GLOBAL* props = nullptr /* kProps: u16[6][prop_stride] */, uint64_t prop_stride = 0, uint32_t role_shift = 0) {
  constexpr uint32_t kFront = Cfg::kFront;
  typedef typename Cfg::CountT CountT;
  constexpr uint32_t kRing = Cfg::kRing, kNear = Cfg::kNear, kCounts = Cfg::kCounts;
  // (tables in LDS were tried for small pages -- u16 entries, 64 KB: one page then fills a CU, and a page on which stage D is the whole cost
  //  gets a twelfth of the throughput of sixteen one-wave pages: 144 ms instead of 20 for the 8192 trial pages of f64 decimals)
  EncChunk PCO_GLOBAL* ch = (EncChunk PCO_GLOBAL*)ws.chunks + t;
  // role_shift: a workgroup's waves are dealt to the SIMDs in order, so with every block's wave k in the same role all four pages of a CU
  // have their stage D on one SIMD and their stage C on another (a step cost four D's worth of issue whatever the other SIMDs did); the
  // blocks rotate the roles instead -- each SIMD gets one wave of every stage
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = (uni(tid >> 6) + role_shift) % Cfg::kWaves;
  const uint32_t wlog = uni(ch->window_n_log), state_n = 1u << uni(ch->state_n_log);
  const uint32_t window_n = 1u << wlog, hash_table_n = 2u << wlog, hash_mask = hash_table_n - 1;
  const uint32_t n = (uint32_t)uni((uint64_t)pg->n); const uint64_t pstart = uni((uint64_t)pg->start);
  const L PCO_GLOBAL* pre = sort_ptr<L>(ws, t, 0) + pstart;
  uint32_t PCO_GLOBAL* lbs = lat_ptr<uint32_t>(ws, t, 0) + pstart;
  L PCO_GLOBAL* out = lat_ptr<L>(ws, t, 1) + pstart;
  uint8_t PCO_LDS* smem = enc_lds_base();
  CountT PCO_LDS* lcounts = (CountT PCO_LDS*)(smem + Cfg::kOffCounts);
  uint64_t PCO_LDS* ring = (uint64_t PCO_LDS*)(smem + Cfg::kOffRing);
  uint16_t PCO_LDS* q_plb = (uint16_t PCO_LDS*)(smem + Cfg::kOffPlb);
  uint8_t PCO_LDS* q_lz = (uint8_t PCO_LDS*)(smem + Cfg::kOffLz);
  uint32_t PCO_LDS* q_lb = (uint32_t PCO_LDS*)(smem + Cfg::kOffLb);
  uint32_t PCO_LDS* abort_flag = q_lb + 2 * 64;   // (one word behind the lookback queue)
  uint32_t PCO_LDS* q_grp = (uint32_t PCO_LDS*)(smem + Cfg::kOffGrp);
  // delta state = the first state_n latents, right aligned (lookback.rs:179-181); state_n == 1 from this encoder
  if (tid == 0) for (uint32_t i = 0; i < state_n && i < 8; i++) pg->moments[i] = i < n ? (uint64_t)pre[i] : 0ull;
  if (n <= state_n) return false;   // (uniform over the block)
  if (tid == 0) *abort_flag = 0;
  for (uint32_t i = tid; i < state_n; i += Cfg::kThreads) ring[i & (kRing - 1)] = (uint64_t)pre[i];   // the positions before the first tile
  const uint32_t n_counts = window_n < n ? window_n : n;
  for (uint32_t i = tid; i < kCounts; i += Cfg::kThreads) lcounts[i] = (CountT)1;
  for (uint32_t i = kCounts + tid; i < n_counts; i += Cfg::kThreads) gcounts[i] = 1;
  if constexpr (!Cfg::kProps) for (uint32_t i = tid; i < hash_table_n / 2; i += Cfg::kThreads) ((uint64_t PCO_GLOBAL*)hash_tbl)[i] = 0ull;   // 2 tables x hash_table_n u16
  __threadfence_block();
  __syncthreads();
  const uint32_t n_tiles = (n - state_n + 63) / 64;
  auto hash_fn = [&](uint64_t x) { x = (x ^ (x >> 32)) * 11400714819323197441ull; x = x ^ (x >> 32); return (uint32_t)x & hash_mask; };
  auto lz_of = [&](L l, L other) { const L d1 = (L)(l - other), d2 = (L)(other - l); const L dlt = d1 < d2 ? d1 : d2; return LBits<L>::v - bitlen<L>(dlt); };
  // the latent `lb` positions before position i (i in the tile a stage is working on): the ring serves the recent ones
  auto latent_back = [&](uint32_t i, uint32_t lb) { return lb < kNear ? (L)ring[(i - lb) & (kRing - 1)] : pre[i - lb]; };
  auto tile_latent = [&](uint32_t i0t) { return i0t < n && lane < n - i0t ? (uint64_t)pre[i0t + lane] : 0ull; };

  // ------------------------------------------------------------------ per-stage state (each wave uses its own part)
  // H: the next two tiles' latents and (HBM tables) the next tile's three table entries, in flight
  uint64_t h_lv = 0, h_lv2 = 0; uint32_t h_val[3] = {0, 0, 0};
  // D: choose_lookbacks' running state (wave-uniform) -- the current best lookback and its count, the four "repeating" proposals and theirs
  uint32_t proposed = 1, best_lookback = 1, repeating_idx = 0;
  uint32_t ring_lb0 = 1, ring_lb1 = 1, ring_lb2 = 1, ring_lb3 = 1, ring_c0 = 1, ring_c1 = 1, ring_c2 = 1, ring_c3 = 1, cnt_best = 1;
  // A: the ranges of the delta'd primary and of the lookbacks
  L mn1 = (L)~(L)0, mx1 = 0; uint32_t mn0 = 0xffffffffu, mx0 = 0;
  uint32_t h_pp[6] = {0, 0, 0, 0, 0, 0};   // (loader) the next tile's six proposals, in flight
  auto tile_props = [&](uint32_t i0t, uint32_t (&pp)[6]) {
    const bool a = i0t < n && lane < n - i0t;
#pragma unroll
    for (int r = 0; r < 6; r++) pp[r] = a ? (uint32_t)props[(uint64_t)r * prop_stride + i0t + lane] : 1u;
  };
  if (wave < kFront) {
    h_lv = tile_latent(state_n); h_lv2 = tile_latent(state_n + 64);
    if constexpr (Cfg::kProps) tile_props(state_n, h_pp);
    else {
      const uint32_t c = wave;
      const uint64_t bucket = h_lv >> (c == 0 ? 0 : 8);
      const bool a = lane < n - state_n;
      const uint32_t s0 = c * hash_table_n + hash_fn(bucket - 1), s1 = c * hash_table_n + hash_fn(bucket), s2 = c * hash_table_n + hash_fn(bucket + 1);
      h_val[0] = a ? (uint32_t)__hip_atomic_load(&hash_tbl[s0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
      h_val[1] = a ? (uint32_t)__hip_atomic_load(&hash_tbl[s1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
      h_val[2] = a ? (uint32_t)__hip_atomic_load(&hash_tbl[s2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    }
  }
  if constexpr (Cfg::kProps) { if (wave == kFront + 1) __builtin_amdgcn_s_setprio(3); else if (wave == kFront) __builtin_amdgcn_s_setprio(1); }   // stage D's chain sets the step: it issues first wherever it shares a SIMD
  uint32_t next_sweep = kLbSweepPeriod;   // (H waves)
  uint32_t d_rounds = 0;                  // (D wave) rounds over the page's first tiles
  uint32_t fast_from = kLbSeqTiles + 1;   // (D wave, kFastD) the first tile that may take the ready-made group maxima: two tiles behind the last count that crossed a power of two
  if (wave == kFront + 1 && lane < 16) proposed = (lane + 1) < state_n ? (lane + 1) : state_n;
#ifdef PCO_LBP_TIMING
  unsigned long long tm_acc = 0, tm_rounds = 0, tm_d[5] = {0, 0, 0, 0, 0}, tm_x = 0, tm_bar = 0;
#define LBP_STAMP(i) do { __asm__ volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long _n = __builtin_readcyclecounter(); tm_d[i] += _n - tm_x; tm_x = _n; } while (0)
#else
#define LBP_STAMP(i) do { } while (0)
#endif

  for (uint32_t step = 0; step < n_tiles + 3; step++) {
#ifdef PCO_LBP_TIMING
    const unsigned long long tm0 = __builtin_readcyclecounter();
#endif
    if (Cfg::kProps && wave < kFront) {
      // ============================================================ L: the latents and the six hash proposals of tile `step`, from their streams
      if (step < n_tiles) {
        const uint32_t i0 = state_n + 64 * step, tile_n = n - i0 < 64 ? n - i0 : 64, ie = i0 + lane;
        const bool act = lane < tile_n;
        if (act) ring[ie & (kRing - 1)] = h_lv;
        uint16_t PCO_LDS* qp = q_plb + (step % 3u) * 6 * 64;
#pragma unroll
        for (int r = 0; r < 6; r++) qp[r * 64 + lane] = (uint16_t)(act ? h_pp[r] : 1u);
        h_lv = h_lv2; h_lv2 = tile_latent(i0 + 128);
        tile_props(i0 + 64, h_pp);
      }
      if constexpr (Cfg::kFastD) {
        // ... and, for tile step - 1 (stage C's tile), the brute-force proposals 1..6: leading-zero counts for stage D's full evaluation, and
        // the best of the six as the counts stand (see stage D: exact whenever no count has crossed a power of two since)
        if (step >= 1 && step - 1 < n_tiles) {
          const uint32_t ts = step - 1, i0 = state_n + 64 * ts, tile_n = n - i0 < 64 ? n - i0 : 64;
          const bool act = lane < tile_n;
          const uint32_t ie = act ? i0 + lane : i0;
          uint8_t PCO_LDS* ql = q_lz + (ts & 1u) * 12 * 64;
          const L l = (L)ring[ie & (kRing - 1)];
          uint32_t best_g = 0, best = 0;
#pragma unroll
          for (uint32_t k = 0; k < 6; k++) {
            const uint32_t b = k + 1 <= ie ? k + 1 : ie;
            const uint32_t lz = lz_of(l, (L)ring[(ie - b) & (kRing - 1)]);
            ql[k * 64 + lane] = (uint8_t)lz;
            const uint32_t g = (32u - clz_u32((uint32_t)lcounts[k])) + lz;
            if (g > best_g) { best_g = g; best = k + 1; }
          }
          q_grp[(ts & 1u) * 128 + lane] = best_g | (best << 8);
        }
      }
    } else if (wave < kFront) {
      // ============================================================ H0 / H1: hash proposals of tile `step` from table `wave`
      const uint32_t c = wave;
      if (step < n_tiles) {
        const uint32_t i0 = state_n + 64 * step, tile_n = n - i0 < 64 ? n - i0 : 64, ie = i0 + lane;
        const bool act = lane < tile_n;
        const uint64_t lv = h_lv;
        if (c == 0 && act) ring[ie & (kRing - 1)] = lv;
        {
          if (i0 >= next_sweep) {
            // sweep of this wave's table: every entry older than the window becomes "window_n + 1 positions old" (stale either way), so that
            // no entry's age can reach 2^16 before the next sweep.  (The entries prefetched for this tile were read before the sweep: a
            // stale one is stale in both forms.)  8-byte L2-served reads: the table's lines were written two bytes at a time by this wave.
            const uint32_t T = i0 & 0xffffu, marker = (i0 - window_n - 1) & 0xffffu;
            uint64_t PCO_GLOBAL* tv = (uint64_t PCO_GLOBAL*)(hash_tbl + (uint64_t)c * hash_table_n);
            for (uint32_t v = lane; v < hash_table_n / 4; v += 64) {
              uint64_t q = __hip_atomic_load(&tv[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              bool changed = false;
#pragma unroll
              for (int k = 0; k < 4; k++) {
                const uint32_t e = (uint32_t)(q >> (16 * k)) & 0xffffu;
                if (((T - e) & 0xffffu) > window_n) { q = (q & ~(0xffffull << (16 * k))) | ((uint64_t)marker << (16 * k)); changed = true; }
              }
              if (changed) tv[v] = q;
            }
            next_sweep += kLbSweepPeriod;
          }
        }
        const uint64_t bucket = lv >> (c == 0 ? 0 : 8);
        uint32_t slot[3], val[3];
        slot[0] = c * hash_table_n + hash_fn(bucket - 1); slot[1] = c * hash_table_n + hash_fn(bucket); slot[2] = c * hash_table_n + hash_fn(bucket + 1);
        for (int r = 0; r < 3; r++) val[r] = h_val[r];
        // (u16 entries: positions mod 2^16; an in-tile hit below stores the hit's position the same way)
        // in-tile hazards: an earlier element of the tile wrote its centre bucket (slot[1]) before we read; each of my three slots needs the
        // LAST earlier lane whose centre slot equals it.  Eight wave votes give every lane the lanes whose centre slot agrees with a
        // slot of mine in its low 8 bits (usually nobody); the few candidates are checked newest first.
        uint32_t jmatch = 0xffffffffu;   // the earlier lane with my own centre slot, if any (then that lane's table update is dead)
        {
          uint64_t vote[8];
#pragma unroll
          for (int b = 0; b < 8; b++) vote[b] = __ballot(act && ((slot[1] >> b) & 1u));
          const uint64_t earlier = __ballot(act) & (((uint64_t)1 << lane) - 1);
          uint64_t cand[3];
#pragma unroll
          for (int r = 0; r < 3; r++) {
            uint64_t m = earlier;
#pragma unroll
            for (int b = 0; b < 8; b++) m &= ((slot[r] >> b) & 1u) ? vote[b] : ~vote[b];
            cand[r] = act ? m : 0ull;
          }
          for (;;) {
            if (!__any(cand[0] != 0 || cand[1] != 0 || cand[2] != 0)) break;
#pragma unroll
            for (int r = 0; r < 3; r++) {
              const uint32_t j = cand[r] ? 63u - (uint32_t)__builtin_clzll(cand[r]) : 0u;
              const uint32_t theirs = (uint32_t)__shfl((int)slot[1], (int)j, 64);   // (every lane takes part in the exchange)
              if (cand[r]) {
                if (theirs == slot[r]) { val[r] = (i0 + j) & 0xffffu; cand[r] = 0; if (r == 1) jmatch = j; }
                else cand[r] &= ~((uint64_t)1 << j);
              }
            }
          }
        }
        {
          // plain stores: of the lanes that share a centre slot only the last may write (lookback.rs:60 in element order)
          uint64_t mm = __ballot(act && jmatch != 0xffffffffu);
          bool shadowed = false;
          while (mm) { const uint32_t k = (uint32_t)__builtin_ctzll(mm); mm &= mm - 1; if ((uint32_t)__builtin_amdgcn_readlane((int)jmatch, (int)k) == lane) shadowed = true; }
          if (act && !shadowed) hash_tbl[slot[1]] = (uint16_t)ie;
        }
        // the next tile's latents and table entries travel while this step's other stages run
        h_lv = h_lv2; h_lv2 = tile_latent(i0 + 128);
        {
          const uint32_t i1 = i0 + 64; const bool a1 = i1 < n && lane < n - i1;
          const uint64_t b1 = h_lv >> (c == 0 ? 0 : 8);
          const uint32_t s0 = c * hash_table_n + hash_fn(b1 - 1), s1 = c * hash_table_n + hash_fn(b1), s2 = c * hash_table_n + hash_fn(b1 + 1);
          h_val[0] = a1 ? (uint32_t)__hip_atomic_load(&hash_tbl[s0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;   // L2-served (this wave's own stores are there)
          h_val[1] = a1 ? (uint32_t)__hip_atomic_load(&hash_tbl[s1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
          h_val[2] = a1 ? (uint32_t)__hip_atomic_load(&hash_tbl[s2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        }
        uint16_t PCO_LDS* qp = q_plb + (step % 3u) * 6 * 64;
#pragma unroll
        for (int r = 0; r < 3; r++) {
          const uint32_t lb = (ie - val[r]) & 0xffffu;   // the entry's age (no entry is ever 2^16 positions old: the sweep)
          const uint32_t pidx = 10 + 3 * c + r;
          const uint32_t plb = lb <= window_n ? lb : (pidx < ie ? pidx : ie);   // lookback.rs:52-56
          qp[(3 * c + r) * 64 + lane] = (uint16_t)(act ? plb : 1u);
        }
      }
    } else if (wave == kFront) {
      // ============================================================ C: the decision-independent candidates of tile step - 1
      if (step >= 1 && step - 1 < n_tiles) {
        const uint32_t ts = step - 1, i0 = state_n + 64 * ts, tile_n = n - i0 < 64 ? n - i0 : 64;
        const bool act = lane < tile_n;
        const uint32_t ie = act ? i0 + lane : i0;
        const uint16_t PCO_LDS* qp = q_plb + (ts % 3u) * 6 * 64;
        uint8_t PCO_LDS* ql = q_lz + (ts & 1u) * 12 * 64;
        const L l = (L)ring[ie & (kRing - 1)];
        uint32_t lb[12];
#pragma unroll
        for (int k = 0; k < 6; k++) lb[k] = (uint32_t)k + 1;   // brute force: 1..6 (clamped to the position on the page's first tile by stage D itself)
#pragma unroll
        for (int r = 0; r < 6; r++) lb[6 + r] = (uint32_t)qp[r * 64 + lane];
        constexpr int kFirst = Cfg::kFastD ? 6 : 0;   // (kFastD: the loader wave takes the brute-force half)
        L c_near[12], c_far[12];
#pragma unroll
        for (int k = kFirst; k < 12; k++) {   // both sources are read for every candidate, unconditionally (a per-lane branch around each read serialises them)
          const uint32_t b = lb[k] <= ie ? lb[k] : ie;
          const bool far = b >= kNear;
          c_near[k] = (L)ring[(ie - b) & (kRing - 1)];
          c_far[k] = k < 6 ? (L)0 : pre[ie - (far ? b : 0u)];
        }
        uint32_t best_g = 0, best = 0;
        // kFastD: the counts of hashed proposals beyond the LDS counts come from global memory (stage D's atomics land in the L2; read past the
        // L1).  On seasonal data a fifth of the elements have such a proposal -- the last equal value, a dozen periods back -- so a tile-wide
        // "has a far count" flag sent every tile through stage D's full evaluation; this stage has the time for the round trip (it waits
        // 3 k cycles at the barrier), stage D has not.
        uint32_t cnt_far[6] = {1, 1, 1, 1, 1, 1};
        if constexpr (Cfg::kFastD && kCounts < (1u << 15)) {
          bool far_any = false;
#pragma unroll
          for (int r = 0; r < 6; r++) far_any = far_any || (act && lb[6 + r] - 1 >= kCounts);
          if (__any(far_any)) {
#pragma unroll
            for (int r = 0; r < 6; r++) if (act && lb[6 + r] - 1 >= kCounts) cnt_far[r] = __hip_atomic_load(&gcounts[lb[6 + r] - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
#pragma unroll
        for (int k = kFirst; k < 12; k++) {
          const uint32_t b = lb[k] <= ie ? lb[k] : ie;
          const uint32_t lz = lz_of(l, (k >= 6 && b >= kNear) ? c_far[k] : c_near[k]);
          ql[k * 64 + lane] = (uint8_t)lz;
          if constexpr (Cfg::kFastD) {   // the best of the six hashed proposals as the counts stand
            const bool near_cnt = kCounts >= (1u << 15) || lb[k] - 1 < kCounts;
            const uint32_t cnt = near_cnt ? (uint32_t)lcounts[near_cnt ? lb[k] - 1 : 0u] : cnt_far[k < 6 ? 0 : k - 6];
            const uint32_t g = (32u - clz_u32(cnt)) + lz;
            if (g > best_g) { best_g = g; best = lb[k]; }
          }
        }
        if constexpr (Cfg::kFastD) q_grp[(ts & 1u) * 128 + 64 + lane] = best_g | (best << 8);
      }
    } else if (wave == kFront + 1) {
      // ============================================================ D: the decisions of tile step - 2
      if (step >= 2 && step - 2 < n_tiles) {
        const uint32_t ts = step - 2, i0 = state_n + 64 * ts, tile_n = n - i0 < 64 ? n - i0 : 64;
        const bool act = lane < tile_n;
        const uint32_t ie = i0 + lane;
        const uint16_t PCO_LDS* qp = q_plb + (ts % 3u) * 6 * 64;
        const uint8_t PCO_LDS* ql = q_lz + (ts & 1u) * 12 * 64;
        uint32_t my_lb = 1;   // lane e keeps the lookback chosen for element e of the tile
        auto count_of = [&](uint32_t lb) -> uint32_t {   // lookback_counts[lb - 1] as of now (far ones live in HBM; only this wave touches them)
          if (kCounts >= (1u << 15) || lb - 1 < kCounts) return (uint32_t)lcounts[lb - 1 < kCounts ? lb - 1 : 0u];
          return __hip_atomic_load(&gcounts[lb - 1], __ATOMIC_RELAXED, kLbScope);
        };
        if (ts < kLbSeqTiles) {
          // ---- the page's first tiles, element by element: lanes 0..15 = the 16 proposals, exactly choose_lookbacks' loop.  (The brute-force
          //      slots fill up over the first 16 positions and overwrite the "repeating" slots on the way: lookback.rs:129-130.)
          //      Every lookback here is at most 64 kLbSeqTiles: latents from the ring, counts from LDS. ----
          for (uint32_t e = 0; e < tile_n; e++) {
            const uint32_t i = i0 + e;
            const L l = (L)ring[i & (kRing - 1)];   // uniform
            if (i <= 16) { const uint32_t new_brute = i < 16 ? i : 16; if (lane == new_brute - 1) proposed = new_brute; }
            else if (lane == 15) proposed = 16;       // (slot 15 is a hash slot: rewritten below)
            if (lane >= 10 && lane < 16) proposed = (uint32_t)qp[(lane - 10) * 64 + e];
            uint32_t key = 0;
            if (lane < 16) {
              const uint32_t lb = proposed;
              const L other = (L)ring[(i - lb) & (kRing - 1)];
              const uint32_t cnt = (uint32_t)lcounts[lb - 1];
              const uint32_t goodness = (32u - clz_u32(cnt)) + lz_of(l, other);
              key = (goodness << 4) | (15u - lane);  // max key = max goodness, first proposal on ties (lookback.rs:88-96)
            }
            // arg-max over lanes 0..15 on the DPP network (row 0): after row_shr 1, 2, 4, 8 lane 15 holds the maximum
            { uint32_t o = dpp0<0x111, 0xf>(key); key = o > key ? o : key; o = dpp0<0x112, 0xf>(key); key = o > key ? o : key;
              o = dpp0<0x114, 0xf>(key); key = o > key ? o : key; o = dpp0<0x118, 0xf>(key); key = o > key ? o : key; }
            const uint32_t best_p = 15u - ((uint32_t)__builtin_amdgcn_readlane((int)key, 15) & 15u);
            const uint32_t new_best = (uint32_t)__builtin_amdgcn_readlane((int)proposed, (int)best_p);
            if (new_best != best_lookback) repeating_idx++;
            if (lane == 6 + (repeating_idx & 3u)) proposed = new_best;
            best_lookback = new_best;
            if (lane == e) my_lb = new_best;
            if (lane == 0) lcounts[new_best - 1] = (CountT)(lcounts[new_best - 1] + 1);
            lb_sync();
          }
          // hand the state over to the tile-parallel path (after every sequential tile: the last one's is what counts)
          ring_lb0 = (uint32_t)__builtin_amdgcn_readlane((int)proposed, 6); ring_lb1 = (uint32_t)__builtin_amdgcn_readlane((int)proposed, 7);
          ring_lb2 = (uint32_t)__builtin_amdgcn_readlane((int)proposed, 8); ring_lb3 = (uint32_t)__builtin_amdgcn_readlane((int)proposed, 9);
          ring_c0 = uni(count_of(ring_lb0)); ring_c1 = uni(count_of(ring_lb1)); ring_c2 = uni(count_of(ring_lb2)); ring_c3 = uni(count_of(ring_lb3));
          cnt_best = uni(count_of(best_lookback));
        } else {
          // ---- later tiles: one lane per element, all 64 decided together under the guess that every element repeats the lookback of the
          //      one before it (B).  Under that guess the "repeating" slots never change and only B's count moves, so an element's 16
          //      candidates and their counts are known without waiting for its predecessors.  The first element that decides otherwise
          //      ends the round: everything before it, and its own decision, were made on the true state; the state is brought up to
          //      date and the rest of the tile is decided again.  Exactly choose_lookbacks' sequence (lookback.rs:101-159). ----
          const uint32_t ie_s = act ? ie : i0;
#ifdef PCO_LBP_TIMING
          tm_x = __builtin_readcyclecounter();
#endif
          const L l = (L)ring[ie_s & (kRing - 1)];
          // kFastD: an element's goodness is bitlen(count of the lookback) + leading zeros of the delta, and a count's bit length only moves
          // when it crosses a power of two -- a few hundred times per page on data with few distinct lookbacks.  The stages in front have
          // evaluated the six brute-force and the six hashed proposals of this tile with the counts as they stood one or two tiles ago and
          // left the first maximum of either group; while no count has crossed a power of two since (fast_from), those maxima ARE what the
          // loop below would find, and a round only has to evaluate the four repeating slots and to combine: group order = proposal order
          // (brute force 0..5, repeating 6..9, hashed 10..15), strict > keeps the first maximum (lookback.rs:88-96).  A crossing -- B's
          // virtual increments inside a round included -- sends the rest of the tile and the next tile through the full evaluation.
          uint32_t g_a = 0, lb_a = 0, g_c = 0, lb_c = 0;
          bool fast = false;
          if constexpr (Cfg::kFastD) {
            if (ts >= fast_from) {
              const uint32_t a = q_grp[(ts & 1u) * 128 + lane], c = q_grp[(ts & 1u) * 128 + 64 + lane];
              g_a = a & 255u; lb_a = a >> 8; g_c = c & 255u; lb_c = (c >> 8) & 0xffffu;
              fast = true;
            }
          }
          bool have_full = false;
          uint32_t s_lb[12], s_lz[12], c_far[6];
#pragma unroll
          for (int k = 0; k < 12; k++) { s_lb[k] = 1; s_lz[k] = 0; }
#pragma unroll
          for (int r = 0; r < 6; r++) c_far[r] = 0;
          auto load_full = [&]() {
            have_full = true;
#pragma unroll
            for (int k = 0; k < 6; k++) s_lb[k] = (uint32_t)k + 1;
#pragma unroll
            for (int r = 0; r < 6; r++) s_lb[6 + r] = (uint32_t)qp[r * 64 + lane];
#pragma unroll
            for (int k = 0; k < 12; k++) s_lz[k] = act ? (uint32_t)ql[k * 64 + lane] : 0u;
            // counts of far hashed proposals (beyond the LDS counts): from HBM, as of now (nothing of this tile has touched them: a tile that
            // has far proposals at all is never fast)
            bool far_any = false;
#pragma unroll
            for (int r = 0; r < 6; r++) far_any = far_any || (act && s_lb[6 + r] - 1 >= kCounts);
            if (kCounts < (1u << 15) && __any(far_any)) {
#pragma unroll
              for (int r = 0; r < 6; r++) { const uint32_t lb = s_lb[6 + r]; if (act && lb - 1 >= kCounts) c_far[r] = __hip_atomic_load(&gcounts[lb - 1], __ATOMIC_RELAXED, kLbScope); }
            }
          };
          if (!fast) load_full();
          LBP_STAMP(0);
#ifdef PCO_LBP_TIMING
          if (fast) tm_d[3]++;
#endif
          // the four repeating slots are the one part of the candidate set that depends on the previous tile's decisions
          uint32_t r_lz0 = act ? lz_of(l, latent_back(ie_s, ring_lb0 <= ie_s ? ring_lb0 : ie_s)) : 0u, r_lz1 = act ? lz_of(l, latent_back(ie_s, ring_lb1 <= ie_s ? ring_lb1 : ie_s)) : 0u;
          uint32_t r_lz2 = act ? lz_of(l, latent_back(ie_s, ring_lb2 <= ie_s ? ring_lb2 : ie_s)) : 0u, r_lz3 = act ? lz_of(l, latent_back(ie_s, ring_lb3 <= ie_s ? ring_lb3 : ie_s)) : 0u;
          LBP_STAMP(1);   // (0: groups / full data loaded; 1: + the repeating slots' latents)
          uint32_t e_start = 0;
          bool crossed = false;   // a count's bit length changed in this tile
          auto add_count = [&](uint32_t lb, uint32_t k) -> uint32_t {   // count `lb` += k for everything that mirrors it; returns the new count
            if (k == 0) return 0u;
            uint32_t now;
            if (kCounts >= (1u << 15) || lb - 1 < kCounts) { const uint32_t idx = lb - 1 < kCounts ? lb - 1 : 0u; now = uni((uint32_t)lcounts[idx]) + k; if (lane == 0) lcounts[idx] = (CountT)now; }
            else {
              uint32_t old = 0; if (lane == 0) old = __hip_atomic_fetch_add(&gcounts[lb - 1], k, __ATOMIC_RELAXED, kLbScope);
              now = uni(old) + k;
#pragma unroll
              for (int r = 0; r < 6; r++) c_far[r] += s_lb[6 + r] == lb ? k : 0u;
            }
            if (clz_u32(now) != clz_u32(now - k)) crossed = true;
            if (ring_lb0 == lb) ring_c0 = now; if (ring_lb1 == lb) ring_c1 = now; if (ring_lb2 == lb) ring_c2 = now; if (ring_lb3 == lb) ring_c3 = now;
            return now;
          };
          for (;;) {
#ifdef PCO_LBP_TIMING
            tm_rounds++;
#endif
            d_rounds++;
            if (d_rounds > kLbAbortRounds && ts < kLbSeqTiles + kLbAbortWindow && n_tiles > 4 * (kLbSeqTiles + kLbAbortWindow)) { if (lane == 0) *abort_flag = 1; break; }
            const uint32_t B = best_lookback;
            const uint32_t cb = cnt_best + (lane - e_start);   // B's count as this element sees it
            uint32_t best_g = 0, best = 0;
            auto consider = [&](uint32_t lb, uint32_t lz, uint32_t cnt) { const uint32_t g = (32u - clz_u32(lb == B ? cb : cnt)) + lz; if (g > best_g) { best_g = g; best = lb; } };
            // (the ready-made maxima hold while no bit length has moved: not in this tile so far, and not by B's increments inside this round)
            if (Cfg::kFastD && fast && !crossed && clz_u32(cnt_best + (tile_n - e_start)) == clz_u32(cnt_best)) {
              best_g = g_a; best = lb_a;
              consider(ring_lb0, r_lz0, ring_c0); consider(ring_lb1, r_lz1, ring_c1); consider(ring_lb2, r_lz2, ring_c2); consider(ring_lb3, r_lz3, ring_c3);
              if (g_c > best_g) { best_g = g_c; best = lb_c; }
            } else {
              if (!have_full) load_full();
#pragma unroll
              for (int k = 0; k < 6; k++) consider(s_lb[k], s_lz[k], (uint32_t)lcounts[k]);
              consider(ring_lb0, r_lz0, ring_c0); consider(ring_lb1, r_lz1, ring_c1); consider(ring_lb2, r_lz2, ring_c2); consider(ring_lb3, r_lz3, ring_c3);
#pragma unroll
              for (int r = 0; r < 6; r++) {
                const uint32_t lb = s_lb[6 + r];
                const bool near_cnt = kCounts >= (1u << 15) || lb - 1 < kCounts;
                const uint32_t nearv = (uint32_t)lcounts[near_cnt ? lb - 1 : 0u];
                consider(lb, s_lz[6 + r], near_cnt ? nearv : c_far[r]);
              }
            }
            const uint64_t mism = __ballot(act && lane >= e_start && best != B);
            const uint32_t e_star = mism ? (uint32_t)__builtin_ctzll(mism) : tile_n;
            if (lane >= e_start && lane < e_star) my_lb = B;
            lb_sync();   // (the counts were read by every lane before lane 0 changes them)
            const uint32_t nb = add_count(B, e_star - e_start);
            if (nb) cnt_best = nb;
            if (e_star >= tile_n) break;
            const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)best, (int)e_star);
            if (lane == e_star) my_lb = c;
            repeating_idx++;
            // the slot takes the new lookback BEFORE its count is bumped; add_count then mirrors the bump into the slot's count
            const uint32_t rs = repeating_idx & 3u;
            const uint32_t c_lz = act && lane > e_star ? lz_of(l, latent_back(ie, c)) : 0u;   // (only the elements still to be decided; c may reach before the page for earlier ones)
            if (rs == 0) { ring_lb0 = c; r_lz0 = c_lz; } else if (rs == 1) { ring_lb1 = c; r_lz1 = c_lz; } else if (rs == 2) { ring_lb2 = c; r_lz2 = c_lz; } else { ring_lb3 = c; r_lz3 = c_lz; }
            cnt_best = add_count(c, 1u);
            best_lookback = c;
            e_start = e_star + 1;
            lb_sync();
            if (e_start >= tile_n) break;
          }
          lb_sync();
          LBP_STAMP(2);   // the rounds
          if (crossed) fast_from = ts + 2;
        }
        q_lb[(ts & 1u) * 64 + lane] = my_lb;
        d_rounds = 0;
      }
    } else {
      // ============================================================ A: lookback.rs:166-185 on tile step - 3
      if (step >= 3 && step - 3 < n_tiles) {
        const uint32_t ts = step - 3, i0 = state_n + 64 * ts, tile_n = n - i0 < 64 ? n - i0 : 64;
        const bool act = lane < tile_n;
        const uint32_t ie = i0 + lane;
        const uint32_t my_lb = q_lb[(ts & 1u) * 64 + lane];
        if (act) {
          const L lv = (L)ring[ie & (kRing - 1)];
          const L other = latent_back(ie, my_lb);
          const L d = (L)(lv - other + lmid<L>());
          out[ie] = d; lbs[ie] = my_lb;
          mn1 = d < mn1 ? d : mn1; mx1 = d > mx1 ? d : mx1; mn0 = my_lb < mn0 ? my_lb : mn0; mx0 = my_lb > mx0 ? my_lb : mx0;
        }
      }
    }
#ifdef PCO_LBP_TIMING
    tm_acc += __builtin_readcyclecounter() - tm0;
    const unsigned long long tm_b0 = __builtin_readcyclecounter();
#endif
    // step barrier: what the stages hand over lives in LDS, so only the LDS queue has to drain -- the global loads a stage sent ahead
    // (next tile's latents and table entries, far latents) and its stores stay in flight across it (__syncthreads would wait for them all)
    __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef PCO_LBP_TIMING
    tm_bar += __builtin_readcyclecounter() - tm_b0;
#endif
    if (uni(*abort_flag)) return true;   // (every wave sees the flag after the same barrier; nothing global but the page's own slices was written)
  }
#ifdef PCO_LBP_TIMING
  if (lane == 0) { atomicAdd(&g_lbp_timing[wave + 2 - kFront], tm_acc); if (wave == kFront + 1) { atomicAdd(&g_lbp_timing[5], tm_rounds); atomicAdd(&g_lbp_timing[6], (unsigned long long)n_tiles); atomicAdd(&g_lbp_timing[7], 1ull); atomicAdd(&g_lbp_timing[8], tm_d[0]); atomicAdd(&g_lbp_timing[9], tm_d[1]); atomicAdd(&g_lbp_timing[10], tm_d[2]); atomicAdd(&g_lbp_timing[11], tm_bar); atomicAdd(&g_lbp_timing[13], tm_d[3]); } if (wave == kFront) atomicAdd(&g_lbp_timing[12], tm_bar); }
#endif
  if (wave == kFront + 2) {
    for (int dlt = 32; dlt >= 1; dlt >>= 1) {
      L o1 = shfl_idx(mn1, (int)(lane ^ dlt)); mn1 = o1 < mn1 ? o1 : mn1;
      L o2 = shfl_idx(mx1, (int)(lane ^ dlt)); mx1 = o2 > mx1 ? o2 : mx1;
      uint32_t o3 = __shfl_xor(mn0, dlt, 64); mn0 = o3 < mn0 ? o3 : mn0;
      uint32_t o4 = __shfl_xor(mx0, dlt, 64); mx0 = o4 > mx0 ? o4 : mx0;
    }
    if (lane == 0) {
      atomicMin((unsigned long long*)&ws.chunks[t].v[1].minv, (unsigned long long)mn1); atomicMax((unsigned long long*)&ws.chunks[t].v[1].maxv, (unsigned long long)mx1);
      atomicMin((unsigned long long*)&ws.chunks[t].v[0].minv, (unsigned long long)mn0); atomicMax((unsigned long long*)&ws.chunks[t].v[0].maxv, (unsigned long long)mx0);
    }
  }
  return false;
}

// grid = a pool of page slots (at most as many as keep their tables in the memory-side cache), one workgroup of five waves per slot; a
// slot takes pages blockIdx.x, blockIdx.x + gridDim.x, ... of the lookback pages (page_ids lists them).  redo[k] = 1: lookback page k
// was handed back to enc_lookback_kernel.
template <class Cfg>
__global__ __launch_bounds__(Cfg::kThreads, Cfg::kProps ? 4 : 1) void enc_lookback_pipe_kernel(EncWorkspace ws, const uint32_t* page_ids, uint32_t n_lb_pages, uint32_t* lb_scratch, uint64_t scratch_stride_u32, uint32_t* redo,
                                                                          const uint16_t* props = nullptr, uint64_t prop_stride = 0, uint32_t role_rotate = 0, const uint32_t* skip = nullptr) {
  uint32_t PCO_GLOBAL* base = (uint32_t PCO_GLOBAL*)lb_scratch + (uint64_t)blockIdx.x * scratch_stride_u32;
  for (uint32_t k = blockIdx.x; k < n_lb_pages; k += gridDim.x) {
    const uint32_t p = page_ids[k];
    EncPage PCO_GLOBAL* pg = (EncPage PCO_GLOBAL*)ws.pages + p;
    if (skip != nullptr && uni(skip[k]) != 0) { if (threadIdx.x == 0) redo[k] = 1; continue; }   // (the pre-pass's screen: a page of enc_lookback_seq_kernel, or -- without it -- of the one-wave kernel)
    if (threadIdx.x == 0) redo[k] = 0;
    if (uni(pg->flags) & kPageFlagMetaOnly) continue;
    const uint32_t t = uni(pg->chunk);
    EncChunk PCO_GLOBAL* ch = (EncChunk PCO_GLOBAL*)ws.chunks + t;
    if (uni(ch->status) != PCO_GFX_OK || uni(ch->delta_kind) != kDeltaLookback) continue;
    const uint32_t wlog = uni(ch->window_n_log);
    uint16_t PCO_GLOBAL* hash_tbl = (uint16_t PCO_GLOBAL*)base; uint32_t PCO_GLOBAL* gcounts = base + (2ull << wlog);   // u16[2][2 << wlog], then u32[1 << wlog]
    const int bits = dtype_bits(uni(ch->dtype));
    const uint16_t PCO_GLOBAL* pk = (const uint16_t PCO_GLOBAL*)props + (uint64_t)k * 6 * prop_stride;   // (kProps: lookback page k's six proposal streams)
    bool aborted;
    if (bits == 64) aborted = lookback_page_pipe<uint64_t, Cfg>(ws, t, pg, hash_tbl, gcounts, pk, prop_stride, role_rotate ? blockIdx.x % Cfg::kWaves : 0u);
    else if (bits == 32) aborted = lookback_page_pipe<uint32_t, Cfg>(ws, t, pg, hash_tbl, gcounts, pk, prop_stride, role_rotate ? blockIdx.x % Cfg::kWaves : 0u);
    else if (bits == 16) aborted = lookback_page_pipe<uint16_t, Cfg>(ws, t, pg, hash_tbl, gcounts, pk, prop_stride, role_rotate ? blockIdx.x % Cfg::kWaves : 0u);
    else aborted = lookback_page_pipe<uint8_t, Cfg>(ws, t, pg, hash_tbl, gcounts, pk, prop_stride, role_rotate ? blockIdx.x % Cfg::kWaves : 0u);
    if (aborted && threadIdx.x == 0) redo[k] = 1;
    __syncthreads();   // the next page re-initialises the LDS every wave of this one may still be reading
  }
}

// =========================================================================================================
// The hash proposals as a pre-pass (round 5).  Slots 10..15 of an element's proposals (lookback.rs:22-64) depend on nothing but the
// latents and the two last-index tables' own earlier updates, and ONE table of a 2^18-number page is 2^16 u16 entries = 128 KB: it fits
// the LDS of a CU.  One workgroup per (page, table): the table lives in LDS for the whole page, the latents are streamed once, and three
// u16 proposals per element and table leave as coalesced streams (12 B per element for both tables, where the tables in HBM cost ~450 B
// of 64-byte-line traffic per element: profiles/r04_c4_pmc_hbm_traffic.txt).  The pipeline above then reads them like the latents.
//
// Only the table accesses are ordered by element; everything else is a function of the tile.  Fifteen worker waves and one sequencer wave
// advance in lockstep, fifteen tiles per step (16 waves: four per SIMD -- with eight, two per SIMD, a step took 6.3 k cycles and the launch
// 49 ms per 4096 pages):
//   worker w, step s     S1 on tile 15 s + w: latents -> the three slots (hash of bucket - 1, bucket, bucket + 1), and the tile's OWN
//                        hazards by eight wave votes (for each of my slots the last earlier lane of the tile whose centre slot is the
//                        same; whether a later lane writes my centre slot) -- into the step's queue
//   sequencer, step s    S2 on the fifteen tiles of step s - 1, in order: three table reads per lane, the in-tile hits put in their
//                        place, the centre slot written by the lanes no later lane shadows.  LDS operations of one wave execute in the
//                        order they are issued, so five tiles' reads and writes at a time are sent back to back without a wait between them
//   worker w, step s     S3 on its tile of step s - 2: entry -> age -> proposal (lookback.rs:50-54), stored to the page's streams
// Entries are positions mod 2^16 with the pipeline's sweep (every 2^14 positions every entry older than the window becomes "window + 1
// positions old"), done by the whole block at a step boundary.
// =========================================================================================================
constexpr uint32_t kLhWorkers = 15, kLhWaves = kLhWorkers + 1, kLhThreads = 64 * kLhWaves, kLhChunk = 5;   // (the sequencer takes the step's tiles five at a time)
static_assert(kLhWorkers % kLhChunk == 0, "whole chunks");
constexpr uint32_t kLhQTile = 3 * 64 * 4;                               // per lane: slot0 | slot1 << 16, slot2 | flags << 16, the in-tile hits -- S2 leaves entry0 | entry1 << 16, entry2 in the first two
// LDS: the table, u16[2 << window_n_log] (128 KB for a full page: one block per CU; 32 KB for the Auto-delta trial samples: two), then the queue
__host__ __device__ constexpr uint32_t lh_queue_off(uint32_t wlog_max) { return 4u << wlog_max; }
__host__ __device__ constexpr uint32_t lh_lds_bytes(uint32_t wlog_max) { return lh_queue_off(wlog_max) + 2 * kLhWorkers * kLhQTile; }   // 154 112 B at window_n_log 15
constexpr uint32_t kLhHas = 7u, kLhShadowed = 8u, kLhAct = 16u;          // flags: has[r] = 1 << r

template <class L>
__device__ bool lookback_hash_page(const EncWorkspace& ws, uint32_t t, const EncPage PCO_GLOBAL* pg, uint32_t c, uint16_t PCO_GLOBAL* props, uint64_t prop_stride, uint32_t queue_off) {   // -> the page goes to the one-wave kernel instead
  const EncChunk PCO_GLOBAL* ch = (const EncChunk PCO_GLOBAL*)ws.chunks + t;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = uni(tid >> 6);
  const uint32_t wlog = uni(ch->window_n_log), state_n = 1u << uni(ch->state_n_log);
  const uint32_t window_n = 1u << wlog, hash_table_n = 2u << wlog, hash_mask = hash_table_n - 1;
  const uint32_t n = (uint32_t)uni((uint64_t)pg->n); const uint64_t pstart = uni((uint64_t)pg->start);
  if (n <= state_n) return false;
  const L PCO_GLOBAL* pre = sort_ptr<L>(ws, t, 0) + pstart;
  uint8_t PCO_LDS* smem = enc_lds_base();
  uint16_t PCO_LDS* tbl = (uint16_t PCO_LDS*)smem;
  bool screened = false;
  if (n <= kLbPipeSmallMaxPage) {
    // A screen for the Auto-delta trial samples: latents that span fewer than 4 n values (the float-mult multiples of decimal data: small random
    // integers) repeat exactly at ever-changing distances, nearly every element picks a lookback nobody picked before, and stage D's speculation
    // fails every round (49 rounds a tile): the pipeline leaves those pages alone (instead of handing them back after sixteen tiles, 3.5 ms per
    // 8192 pages for nothing) and enc_lookback_seq_kernel decides them element by element.  Which kernel takes a page changes no byte.
    typedef typename std::conditional<sizeof(L) == 8, uint64_t, uint32_t>::type W;
    W mn = (W)(L)~(L)0, mx = 0;
    for (uint32_t i = tid; i < n; i += kLhThreads) { const W x = (W)pre[i]; mn = x < mn ? x : mn; mx = x > mx ? x : mx; }
    mn = wave_butterfly(mn, [](W a, W b) { return a < b ? a : b; }); mx = wave_butterfly(mx, [](W a, W b) { return a > b ? a : b; });
    uint64_t PCO_LDS* red = (uint64_t PCO_LDS*)smem;
    if (lane == 0) { red[2 * wave] = (uint64_t)mn; red[2 * wave + 1] = (uint64_t)mx; }
    __syncthreads();
    uint64_t lo = ~0ull, hi = 0;
    for (uint32_t w = 0; w < kLhWaves; w++) { const uint64_t a = red[2 * w], b = red[2 * w + 1]; lo = a < lo ? a : lo; hi = b > hi ? b : hi; }
    __syncthreads();
    screened = hi - lo < 4ull * n;   // (the proposals are computed all the same: enc_lookback_seq_kernel reads them)
  }
  for (uint32_t i = tid; i < hash_table_n / 4; i += kLhThreads) ((uint64_t PCO_LDS*)tbl)[i] = 0ull;
  __syncthreads();
  const uint32_t n_tiles = (n - state_n + 63) / 64, n_steps = (n_tiles + kLhWorkers - 1) / kLhWorkers + 2;
  auto hash_fn = [&](uint64_t x) { x = (x ^ (x >> 32)) * 11400714819323197441ull; x = x ^ (x >> 32); return (uint32_t)x & hash_mask; };
  auto tile_latent = [&](uint32_t tile) { const uint32_t i0t = state_n + 64 * tile; return tile < n_tiles && lane < n - i0t ? (uint64_t)pre[i0t + lane] : 0ull; };
  auto queue = [&](uint32_t parity, uint32_t w) { return (uint32_t PCO_LDS*)(smem + queue_off + (parity * kLhWorkers + w) * kLhQTile); };
  uint64_t lv_next = wave < kLhWorkers ? tile_latent(wave) : 0ull;
  uint32_t next_sweep = kLbSweepPeriod;
  // The sequencer's 500 instructions per step are the block's critical path, and it shares its SIMD with three workers: at equal priority it got
  // a quarter of the issue slots (7.0 k busy cycles per step against the workers' 3.8 k)
  if (wave == kLhWorkers) __builtin_amdgcn_s_setprio(3);
#ifdef PCO_LBP_TIMING
  unsigned long long th_busy = 0;
#endif
  for (uint32_t step = 0; step < n_steps; step++) {
#ifdef PCO_LBP_TIMING
    const unsigned long long th0 = __builtin_readcyclecounter();
#endif
    // ---- the sweep, by everybody, when the sequencer's next tile has passed the mark (it keeps every age below 2^16; when exactly it
    //      happens changes no proposal: an entry it rewrites is stale before and after) ----
    const uint32_t p2 = state_n + 64 * (step > 0 ? (step - 1) * kLhWorkers : 0u);   // position of the sequencer's first tile of this step
    if (p2 >= next_sweep) {
      const uint32_t T = p2 & 0xffffu, marker = (p2 - window_n - 1) & 0xffffu;
      for (uint32_t v = tid; v < hash_table_n / 4; v += kLhThreads) {
        uint64_t q = ((uint64_t PCO_LDS*)tbl)[v];
        bool changed = false;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const uint32_t e = (uint32_t)(q >> (16 * k)) & 0xffffu;
          if (((T - e) & 0xffffu) > window_n) { q = (q & ~(0xffffull << (16 * k))) | ((uint64_t)marker << (16 * k)); changed = true; }
        }
        if (changed) ((uint64_t PCO_LDS*)tbl)[v] = q;
      }
      next_sweep += kLbSweepPeriod;
      __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (wave < kLhWorkers) {
      // ---- S3: the tile this wave prepared two steps ago; its entries are in the queue buffer S1 is about to reuse ----
      if (step >= 2) {
        const uint32_t tile = (step - 2) * kLhWorkers + wave;
        if (tile < n_tiles) {
          const uint32_t PCO_LDS* q = queue(step & 1u, wave);
          const uint32_t i0 = state_n + 64 * tile, ie = i0 + lane;
          const uint32_t e01 = q[lane], e2 = q[64 + lane], hz = q[128 + lane], fl = e2 >> 16;
          if (lane < n - i0) {
#pragma unroll
            for (uint32_t r = 0; r < 3; r++) {
              uint32_t val = r == 0 ? (e01 & 0xffffu) : (r == 1 ? (e01 >> 16) : (e2 & 0xffffu));
              if ((fl >> r) & 1u) val = (i0 + ((hz >> (6 * r)) & 63u)) & 0xffffu;   // an earlier lane of this tile had written the slot: its position, not the table's entry
              const uint32_t lb = (ie - val) & 0xffffu;   // the entry's age (no entry is ever 2^16 positions old: the sweep)
              const uint32_t pidx = 10 + 3 * c + r;
              const uint32_t plb = lb <= window_n ? lb : (pidx < ie ? pidx : ie);   // lookback.rs:50-54
              props[(uint64_t)(3 * c + r) * prop_stride + ie] = (uint16_t)plb;
            }
          }
        }
      }
      // ---- S1: slots and in-tile hazards of tile step * W + wave ----
      const uint32_t tile = step * kLhWorkers + wave;
      if (tile < n_tiles) {
        const uint32_t i0 = state_n + 64 * tile, tile_n = n - i0 < 64 ? n - i0 : 64;
        const bool act = lane < tile_n;
        const uint64_t bucket = lv_next >> (c == 0 ? 0 : 8);
        lv_next = tile_latent(tile + kLhWorkers);   // (travels across the barrier)
        uint32_t slot[3];
        slot[0] = hash_fn(bucket - 1); slot[1] = hash_fn(bucket); slot[2] = hash_fn(bucket + 1);
        // for each of my three slots the LAST earlier lane whose centre slot equals it: eight votes give the lanes whose centre slot agrees
        // with a slot of mine in its low 8 bits (usually nobody); the few candidates are checked newest first
        uint32_t hit[3] = {0, 0, 0}, flags = act ? kLhAct : 0u;
        {
          uint64_t vote[8];
#pragma unroll
          for (int b = 0; b < 8; b++) vote[b] = __ballot(((slot[1] >> b) & 1u) != 0);   // (lanes beyond the tile: masked out by `earlier`)
          const uint64_t earlier = __ballot(act) & (((uint64_t)1 << lane) - 1);
          // (a lane j differs from my slot r in bit b iff vote_b's bit j differs from that bit of mine: the differences OR'd over the eight
          //  bits, one three-input bit operation per half and bit -- as selects and ANDs of 64-bit masks this block was 250 instructions)
          uint64_t cand[3];
#pragma unroll
          for (int r = 0; r < 3; r++) {
            uint32_t dlo = 0, dhi = 0;
#pragma unroll
            for (int b = 0; b < 8; b++) {
              const uint32_t ext = (uint32_t)((int32_t)(slot[r] << (31 - b)) >> 31);   // my bit b, on every bit position
              dlo |= (uint32_t)vote[b] ^ ext; dhi |= (uint32_t)(vote[b] >> 32) ^ ext;
            }
            cand[r] = act ? earlier & ~(((uint64_t)dhi << 32) | dlo) : 0ull;
          }
          for (;;) {
            if (!__any(cand[0] != 0 || cand[1] != 0 || cand[2] != 0)) break;
#pragma unroll
            for (int r = 0; r < 3; r++) {
              const uint32_t j = cand[r] ? 63u - (uint32_t)__builtin_clzll(cand[r]) : 0u;
              const uint32_t theirs = (uint32_t)__shfl((int)slot[1], (int)j, 64);   // (every lane takes part in the exchange)
              if (cand[r]) {
                if (theirs == slot[r]) { hit[r] = j; flags |= 1u << r; cand[r] = 0; }
                else cand[r] &= ~((uint64_t)1 << j);
              }
            }
          }
        }
        // of the lanes that share a centre slot only the last may write (lookback.rs:57-62 in element order): I am shadowed if a later lane's
        // centre-slot hit is me
        {
          uint64_t mm = __ballot((flags & 2u) != 0);
          while (mm) { const uint32_t k = (uint32_t)__builtin_ctzll(mm); mm &= mm - 1; if ((uint32_t)__builtin_amdgcn_readlane((int)hit[1], (int)k) == lane) flags |= kLhShadowed; }
        }
        uint32_t PCO_LDS* q = queue(step & 1u, wave);
        q[lane] = slot[0] | (slot[1] << 16); q[64 + lane] = slot[2] | (flags << 16); q[128 + lane] = hit[0] | (hit[1] << 6) | (hit[2] << 12);
      }
    } else if (step >= 1) {
      // ---- S2: the table, tile after tile, five tiles' operations in flight (a tile's three reads, then its write, then the next tile's
      //      reads: the LDS executes a wave's operations in the order they were issued, so nothing waits in between) ----
      for (uint32_t chunk = 0; chunk < kLhWorkers / kLhChunk; chunk++) {
        const uint32_t tile0 = (step - 1) * kLhWorkers + chunk * kLhChunk;
        if (tile0 >= n_tiles) break;
        uint32_t d0[kLhChunk], d1[kLhChunk], v0[kLhChunk], v1[kLhChunk], v2[kLhChunk];
#pragma unroll
        for (uint32_t w = 0; w < kLhChunk; w++) {
          const uint32_t PCO_LDS* q = queue((step - 1) & 1u, chunk * kLhChunk + w);
          const bool valid = tile0 + w < n_tiles;   // (uniform)
          d0[w] = valid ? q[lane] : 0u; d1[w] = valid ? q[64 + lane] : 0u;
        }
#pragma unroll
        for (uint32_t w = 0; w < kLhChunk; w++) {
          const uint32_t s0 = d0[w] & 0xffffu, s1 = d0[w] >> 16, s2 = d1[w] & 0xffffu, fl = d1[w] >> 16;
          v0[w] = tbl[s0]; v1[w] = tbl[s1]; v2[w] = tbl[s2];
          if ((fl & (kLhAct | kLhShadowed)) == kLhAct) tbl[s1] = (uint16_t)(state_n + 64 * (tile0 + w) + lane);
        }
        // (the entries as read; the in-tile hits are put in their place by the worker that stores the proposals -- every instruction taken off
        //  this wave is taken off the block's critical path)
#pragma unroll
        for (uint32_t w = 0; w < kLhChunk; w++) {
          if (tile0 + w < n_tiles) {
            uint32_t PCO_LDS* q = queue((step - 1) & 1u, chunk * kLhChunk + w);
            q[lane] = v0[w] | (v1[w] << 16); q[64 + lane] = v2[w] | (d1[w] & 0xffff0000u);
          }
        }
      }
    }
#ifdef PCO_LBP_TIMING
    __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    th_busy += __builtin_readcyclecounter() - th0;
#endif
    __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
#ifdef PCO_LBP_TIMING
  if (lane == 0 && (wave == 0 || wave == kLhWorkers)) { atomicAdd(&g_lbp_timing[wave == 0 ? 14 : 15], th_busy); if (wave == 0) atomicAdd(&g_lbp_timing[0], (unsigned long long)n_steps); }
#endif
  return screened;
}

// grid = 2 x the lookback pages (item = 2 k + table), one block per CU (the table)
__global__ __launch_bounds__(kLhThreads) void enc_lookback_hash_kernel(EncWorkspace ws, const uint32_t* page_ids, uint32_t n_lb_pages, uint16_t* props, uint64_t prop_stride, uint32_t queue_off /* lh_queue_off(the call's largest window_n_log) */,
                                                                     uint32_t* skip /* [page]: 1 = not the pipeline's (see the screen in lookback_hash_page) */) {
  const uint32_t k = blockIdx.x >> 1, c = blockIdx.x & 1u;
  if (k >= n_lb_pages) return;
  if (c == 0 && threadIdx.x == 0) skip[k] = 0u;   // (rewritten below for the pages the screen takes out)
  const EncPage PCO_GLOBAL* pg = (const EncPage PCO_GLOBAL*)ws.pages + page_ids[k];
  if (uni(pg->flags) & kPageFlagMetaOnly) return;
  const uint32_t t = uni(pg->chunk);
  const EncChunk PCO_GLOBAL* ch = (const EncChunk PCO_GLOBAL*)ws.chunks + t;
  if (uni(ch->status) != PCO_GFX_OK || uni(ch->delta_kind) != kDeltaLookback || (4u << uni(ch->window_n_log)) > queue_off) return;
  uint16_t PCO_GLOBAL* pk = (uint16_t PCO_GLOBAL*)props + (uint64_t)k * 6 * prop_stride;
  const int bits = dtype_bits(uni(ch->dtype));
  bool skipped;
  if (bits == 64) skipped = lookback_hash_page<uint64_t>(ws, t, pg, c, pk, prop_stride, queue_off);
  else if (bits == 32) skipped = lookback_hash_page<uint32_t>(ws, t, pg, c, pk, prop_stride, queue_off);
  else if (bits == 16) skipped = lookback_hash_page<uint16_t>(ws, t, pg, c, pk, prop_stride, queue_off);
  else skipped = lookback_hash_page<uint8_t>(ws, t, pg, c, pk, prop_stride, queue_off);
  if (c == 0 && threadIdx.x == 0) skip[k] = skipped ? 1u : 0u;
}

// =========================================================================================================
// The screened pages, element by element (round 5).  A page the pre-pass's screen took out -- at most 8192 latents that span fewer than
// 32768 values -- changes its best lookback at nearly every element, so nothing is gained by deciding 64 elements on a guess: what counts is
// the latency of ONE element's decision.  The one-wave kernel paid two HBM round trips per round on such a page (the count of a lookback
// beyond its 256 LDS counters, the latent at a freshly chosen distance: 9.9 ms a page, 19.7 ms per 8192 trial pages of decimal data).  Here
// everything a decision reads is in LDS: the page's latents as their low 16 bits (two latents of the page differ by less than 2^15, so the
// difference of the low halves, sign-extended, IS the difference), every lookback's count as u16 (a count never exceeds the page), the tile's
// hash proposals from the pre-pass.  Lanes 0..15 hold the sixteen proposals of lookback.rs:101-159 and choose by a DPP arg-max, exactly as
// the one-wave kernel's first tile does; 27 KB of LDS for a 6554-number trial page, five pages per CU.
// =========================================================================================================
constexpr uint32_t kLbSeqMaxPage = 8192;
constexpr uint32_t kLbSeqHpBytes = 2 * 64 * 8 * 2;   // u16[2][64][8]: this tile's and the next one's six hash proposals per element
__host__ __device__ constexpr uint32_t lbseq_round(uint32_t page_max) { return (page_max + 63u) & ~63u; }
__host__ __device__ constexpr uint32_t lbseq_lds_bytes(uint32_t page_max) { return kLbSeqHpBytes + 4u * lbseq_round(page_max); }

template <class L>
__device__ void lookback_seq_page(const EncWorkspace& ws, uint32_t t, EncPage PCO_GLOBAL* pg, const uint16_t PCO_GLOBAL* props, uint64_t prop_stride, uint32_t n_round) {
  const uint32_t lane = lane_id();
  const uint32_t n = (uint32_t)uni((uint64_t)pg->n); const uint64_t pstart = uni((uint64_t)pg->start);
  const L PCO_GLOBAL* pre = sort_ptr<L>(ws, t, 0) + pstart;
  uint32_t PCO_GLOBAL* lbs = lat_ptr<uint32_t>(ws, t, 0) + pstart;
  L PCO_GLOBAL* out = lat_ptr<L>(ws, t, 1) + pstart;
  uint16_t PCO_LDS* hp = (uint16_t PCO_LDS*)enc_lds_base();
  uint16_t PCO_LDS* lat = hp + kLbSeqHpBytes / 2;
  uint16_t PCO_LDS* cnt = lat + n_round;
  if (lane == 0) pg->moments[0] = n ? (uint64_t)pre[0] : 0ull;   // the delta state (lookback.rs:179-181), state_n == 1
  if (n <= 1) return;
  for (uint32_t i = lane; i < n; i += 64) { lat[i] = (uint16_t)pre[i]; cnt[i] = 1; }
  auto tile_props = [&](uint32_t i0t, uint32_t (&pp)[6]) {
    const bool a = i0t < n && lane < n - i0t;
#pragma unroll
    for (int r = 0; r < 6; r++) pp[r] = a ? (uint32_t)props[(uint64_t)r * prop_stride + i0t + lane] : 1u;
  };
  auto hp_store = [&](uint32_t tile, const uint32_t (&pp)[6]) {   // the six proposals of element 1 + 64 tile + lane: row ((tile & 1) * 64 + lane)
    uint64_t PCO_LDS* h8 = (uint64_t PCO_LDS*)(hp + ((tile & 1u) * 64u + lane) * 8u);
    h8[0] = (uint64_t)(pp[0] | (pp[1] << 16)) | ((uint64_t)(pp[2] | (pp[3] << 16)) << 32); h8[1] = (uint64_t)(pp[4] | (pp[5] << 16));
  };
  const uint32_t hl = lane >= 10 && lane < 16 ? lane - 10 : 0u;   // (my column of a row)
  auto hp_row = [&](uint32_t i) { return (uint32_t)hp[((i - 1u) & 127u) * 8u + hl]; };
  uint32_t pf[6]; tile_props(1, pf); hp_store(0, pf); tile_props(65, pf); hp_store(1, pf); tile_props(129, pf);   // (two tiles' rows in LDS, the third in flight)
  uint32_t P = 1, best_lookback = 1, repeating_idx = 0;   // lanes 0..15: proposed_lookbacks[lane] = min(lane + 1, state_n)
  L mn1 = (L)~(L)0, mx1 = 0; uint32_t mn0 = 0xffffffffu, mx0 = 0, my_lb = 1;
  const bool hashed = lane >= 10 && lane < 16;
  auto lz_part = [&](uint32_t l, uint32_t other) {   // leading zeros of |l - other| as an L (lookback.rs:76-80), from the low halves
    const int32_t d = (int32_t)(int16_t)(uint16_t)(l - other);
    uint32_t a = (uint32_t)(d < 0 ? -d : d);
    if (LBits<L>::v == 8) a = a > 128u ? 256u - a : a;   // (the reference takes the smaller of the two WRAPPING differences: an 8-bit page can span more than half its type)
    return LBits<L>::v - (32u - clz_u32(a));
  };
  auto arg_max = [&](uint32_t key) {   // over lanes 0..15 on the DPP network (row 0): after row_shr 1, 2, 4, 8 lane 15 holds the maximum
    uint32_t o = dpp0<0x111, 0xf>(key); key = o > key ? o : key; o = dpp0<0x112, 0xf>(key); key = o > key ? o : key;
    o = dpp0<0x114, 0xf>(key); key = o > key ? o : key; o = dpp0<0x118, 0xf>(key); key = o > key ? o : key;
    return 15u - ((uint32_t)__builtin_amdgcn_readlane((int)key, 15) & 15u);
  };
  auto flush_tile = [&](uint32_t i0, uint32_t tile_n) {   // (lookback.rs:166-185)
    if (lane < tile_n) {
      const uint32_t ie = i0 + lane;
      const int32_t d = (int32_t)(int16_t)(uint16_t)((uint32_t)lat[ie] - (uint32_t)lat[ie - my_lb]);
      const L dl = (L)((L)(int64_t)d + lmid<L>());
      lbs[ie] = my_lb; out[ie] = dl;
      mn0 = my_lb < mn0 ? my_lb : mn0; mx0 = my_lb > mx0 ? my_lb : mx0; mn1 = dl < mn1 ? dl : mn1; mx1 = dl > mx1 ? dl : mx1;
    }
  };
  lb_sync();
  // ---- positions 1..16: the brute-force slots fill up and overwrite the "repeating" slots on the way (lookback.rs:129-130) ----
  const uint32_t n_head = n - 1 < 16 ? n - 1 : 16;
  for (uint32_t i = 1; i <= n_head; i++) {
    const uint32_t l = lat[i];   // uniform
    { const uint32_t new_brute = i < 16 ? i : 16; if (lane == new_brute - 1) P = new_brute; }
    if (hashed) P = hp_row(i);
    const uint32_t key = lane < 16 ? ((((32u - clz_u32((uint32_t)cnt[P - 1])) + lz_part(l, lat[i - P])) << 4) | (15u - lane)) : 0u;   // max key = max goodness, first proposal on ties (lookback.rs:88-96)
    const uint32_t best_p = arg_max(key);
    const uint32_t new_best = (uint32_t)__builtin_amdgcn_readlane((int)P, (int)best_p);
    if (new_best != best_lookback) repeating_idx++;
    if (lane == 6 + (repeating_idx & 3u)) P = new_best;
    best_lookback = new_best;
    if (lane == i - 1) my_lb = new_best;
    if (lane == 0) cnt[new_best - 1] += 1;
    lb_sync();
  }
  if (n - 1 <= 16) flush_tile(1, n - 1);
  else {
    // ---- from position 17 on nothing but the decisions changes a proposal.  One wave per SIMD issues every instruction of the loop back to
    //      back, so the loop is kept SHORT rather than latency-free: the element's latent and hash proposals (which depend on no decision) are
    //      fetched one element ahead, each proposal's latent and count after the previous decision (one LDS round trip), and the winner's
    //      count comes back by readlane instead of a read-modify-write.  (Fetching everything ahead and patching it with the decision --
    //      110 instructions an element instead of 60 -- was slower: 14.9 ms against 13.6 per 8192 trial pages.) ----
    uint32_t l = lat[17], H1 = hp_row(17);
    for (uint32_t i = 17; i < n; i++) {
      const uint32_t e = (i - 1u) & 63u;
      if (e == 0) { hp_store((i >> 6) + 1u, pf); tile_props(i + 128u, pf); lb_sync(); }   // the next tile's rows; the one after it in flight
      if (hashed) P = H1;
      const uint32_t O = lat[i - P], C = cnt[P - 1];
      const uint32_t ln = lat[i + 1], H2 = hp_row(i + 1);
      const uint32_t key = lane < 16 ? ((((32u - clz_u32(C)) + lz_part(l, O)) << 4) | (15u - lane)) : 0u;
      const uint32_t best_p = arg_max(key);
      const uint32_t new_best = (uint32_t)__builtin_amdgcn_readlane((int)P, (int)best_p);
      const uint32_t cw = (uint32_t)__builtin_amdgcn_readlane((int)C, (int)best_p) + 1u;
      if (lane == 0) cnt[new_best - 1] = (uint16_t)cw;
      if (new_best != best_lookback) repeating_idx++;
      best_lookback = new_best;
      if (lane == e) my_lb = new_best;
      if (lane == 6 + (repeating_idx & 3u)) P = new_best;
      l = ln; H1 = H2;
      lb_sync();
      if (e == 63u || i + 1 == n) flush_tile(i - e, e + 1u);
    }
  }
  for (int dlt = 32; dlt >= 1; dlt >>= 1) {
    L o1 = shfl_idx(mn1, (int)(lane ^ dlt)); mn1 = o1 < mn1 ? o1 : mn1;
    L o2 = shfl_idx(mx1, (int)(lane ^ dlt)); mx1 = o2 > mx1 ? o2 : mx1;
    uint32_t o3 = __shfl_xor(mn0, dlt, 64); mn0 = o3 < mn0 ? o3 : mn0;
    uint32_t o4 = __shfl_xor(mx0, dlt, 64); mx0 = o4 > mx0 ? o4 : mx0;
  }
  if (lane == 0) {
    atomicMin((unsigned long long*)&ws.chunks[t].v[1].minv, (unsigned long long)mn1); atomicMax((unsigned long long*)&ws.chunks[t].v[1].maxv, (unsigned long long)mx1);
    atomicMin((unsigned long long*)&ws.chunks[t].v[0].minv, (unsigned long long)mn0); atomicMax((unsigned long long*)&ws.chunks[t].v[0].maxv, (unsigned long long)mx0);
  }
}

// grid = the lookback pages, one wave each; takes the pages the pre-pass's screen marked (skip[k] == 1) and clears their redo flag, so that
// enc_lookback_kernel behind it is left with the pages the pipeline handed back
__global__ __launch_bounds__(64) void enc_lookback_seq_kernel(EncWorkspace ws, const uint32_t* page_ids, uint32_t n_lb_pages, const uint16_t* props, uint64_t prop_stride, const uint32_t* skip, uint32_t* redo, uint32_t n_round) {
  const uint32_t k = blockIdx.x;
  if (k >= n_lb_pages || uni(skip[k]) == 0) return;
  EncPage PCO_GLOBAL* pg = (EncPage PCO_GLOBAL*)ws.pages + page_ids[k];
  if (uni(pg->flags) & kPageFlagMetaOnly) return;
  const uint32_t t = uni(pg->chunk);
  EncChunk PCO_GLOBAL* ch = (EncChunk PCO_GLOBAL*)ws.chunks + t;
  if (uni(ch->status) != PCO_GFX_OK || uni(ch->delta_kind) != kDeltaLookback || uni(ch->state_n_log) != 0) return;
  if ((uint32_t)uni((uint64_t)pg->n) > n_round) return;
  const uint16_t PCO_GLOBAL* pk = (const uint16_t PCO_GLOBAL*)props + (uint64_t)k * 6 * prop_stride;
  const int bits = dtype_bits(uni(ch->dtype));
  if (bits == 64) lookback_seq_page<uint64_t>(ws, t, pg, pk, prop_stride, n_round);
  else if (bits == 32) lookback_seq_page<uint32_t>(ws, t, pg, pk, prop_stride, n_round);
  else if (bits == 16) lookback_seq_page<uint16_t>(ws, t, pg, pk, prop_stride, n_round);
  else lookback_seq_page<uint8_t>(ws, t, pg, pk, prop_stride, n_round);
  if (threadIdx.x == 0) redo[k] = 0;
}

}  // namespace pcogfx
