// pco_gfx.hip -- the C ABI of libpco_gfx.so (see include/pco_gfx.h) and the launch logic.
// One translation unit: the gfx950 kernels are included below.  There is NO CPU codec here: without a
// HIP device every compute entry point fails with PCO_GFX_DEVICE_ERROR.
#include "pco_host.h"
#include "pco_half.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <sched.h>
#include <pthread.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <unordered_map>

#include "pco_auto_host.inc"
#include "decode_kernel.hip"
#include "decode_trail.hip"   // (includes decode_fast.hip, which includes decode_kernel.hip)
#include "encode_kernels.hip"
#include "encode_lookback.hip"
#include "encode_hist_select.hip"
#include "encode_hist_literal.hip"
#include "auto_mode_kernels.hip"
#include "encode_fast.hip"
#include "encode_walkseg.hip"
#include "encode_walkpack.hip"
#include "stream_kernels.hip"

namespace pcogfx {

static thread_local HostError g_err{PCO_GFX_OK, ""};
void set_error(int status, const std::string& msg) { g_err.status = status; g_err.msg = msg; }
void clear_error() { g_err.status = PCO_GFX_OK; g_err.msg.clear(); }

struct WorkspaceHolder {
  std::unordered_map<int, std::unique_ptr<Workspace>> by_device;   // one workspace per (thread, device)
  // A worker thread that exits gives its device and pinned memory back.  The main thread's holder is destroyed during process
  // teardown, when the HIP runtime may already be unloading: there the driver reclaims everything and nothing is touched.
  ~WorkspaceHolder() {
    const bool worker = (long)syscall(SYS_gettid) != (long)getpid();
    for (auto& kv : by_device) {
      if (worker) { int cur = 0; const bool sw = hipGetDevice(&cur) == hipSuccess && cur != kv.first && hipSetDevice(kv.first) == hipSuccess; kv.second->release_all(); if (kv.second->last_event) (void)hipEventDestroy(kv.second->last_event); if (sw) (void)hipSetDevice(cur); }
      else (void)kv.second.release();   // (leaked on purpose, see above)
    }
  }
};
static thread_local WorkspaceHolder g_ws_holder;
Workspace& workspace() {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) throw HostError{PCO_GFX_DEVICE_ERROR, std::string("no HIP device: ") + hipGetErrorString(e)};
  std::unique_ptr<Workspace>& slot = g_ws_holder.by_device[dev];
  if (!slot) { slot.reset(new Workspace()); slot->device = dev; }
  return *slot;
}

static void require_device() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    throw HostError{PCO_GFX_DEVICE_ERROR, "libpco_gfx: no MI355X/HIP device visible; this library has no CPU fallback"};
}

// ---- guarantees (host arithmetic only) ----
static size_t baseline_meta_max_size(int latent_bits) {  // wrapped/guarantee.rs:11-33 + metadata/chunk.rs:105-113
  const size_t delta_max_bits = 4 + 5 + 5 + 64 + 32 * 32;  // DeltaEncoding::MAX_BIT_SIZE
  const size_t var_bits = kBitsAnsSizeLog + kBitsNBins + (0 + (size_t)latent_bits + offset_bits_bits(latent_bits));
  return (kBitsModeVariant + delta_max_bits + var_bits + 7) / 8;
}
size_t guarantee_wrapped_chunk_size(int latent_bits, size_t n) { return baseline_meta_max_size(latent_bits) + (n * (size_t)latent_bits + 7) / 8; }
size_t guarantee_standalone_chunk_size(int latent_bits, size_t n) { return 1 + 3 + guarantee_wrapped_chunk_size(latent_bits, n); }
size_t guarantee_standalone_header_size() { return 4 + 1 + (kBitsVarintPower + 64 + 8 + 7) / 8 + 2; }
bool n_per_page(uint64_t max_page_n, size_t n, std::vector<size_t>& out) {
  out.clear();
  if (max_page_n == 0) max_page_n = 1u << 18;
  if (n == 0) return true;
  size_t n_pages = (n + max_page_n - 1) / max_page_n;
  size_t low = n / n_pages, r = n % n_pages;
  out.assign(n_pages, low);
  for (size_t i = 0; i < r; i++) out[i] = low + 1;
  return true;
}

static PcoError fail_with(const HostError& e, PcoError code) { set_error(e.status, e.msg); return code; }

// ---- optional per-kernel timing with HIP events on the launch stream (used by bench.py's roofline) ----
struct KernelProfile {
  bool on = false;
  struct Rec { const char* name; hipEvent_t a, b; };
  std::vector<Rec> recs;
  std::vector<hipEvent_t> pool;   // events are reused from one profiled region to the next: creating two per launch showed up in host time
  bool take(hipEvent_t& e) { if (!pool.empty()) { e = pool.back(); pool.pop_back(); return true; } return hipEventCreate(&e) == hipSuccess; }
  void give(hipEvent_t e) { pool.push_back(e); }
};
static thread_local KernelProfile g_prof;
struct ScopedKernelTimer {
  hipStream_t s; hipEvent_t b{}; bool live = false;
  ScopedKernelTimer(const char* name, hipStream_t stream) : s(stream) {
    if (!g_prof.on) return;
    hipEvent_t a;
    if (!g_prof.take(a)) return;
    if (!g_prof.take(b)) { g_prof.give(a); return; }
    (void)hipEventRecord(a, s);
    g_prof.recs.push_back({name, a, b}); live = true;
  }
  ~ScopedKernelTimer() { if (live) (void)hipEventRecord(b, s); }
};
#define PCO_TIMED_LAUNCH(name, stream, ...) do { ScopedKernelTimer _t(name, stream); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

// ---------------------------------------------------------------------------------------------------------
// decode launch
// ---------------------------------------------------------------------------------------------------------
static uint32_t g_decode_lds_bytes = 16 * 1024;  // dynamic LDS per wave (fixed area + tANS tables)
static bool g_decode_fast = std::getenv("PCO_GFX_NO_FAST_DECODE") == nullptr;  // A/B switch for the two-kernel path
// The expanders of the common chunks run on a second stream UNDER the walk (decode_trail.hip); PCO_GFX_DEC_TRAIL=0 keeps the two kernels
// back to back (A/B switch).
static bool g_decode_trail = env_not_zero("PCO_GFX_DEC_TRAIL");
// measurement / test switches: 's' = the expanders on the walker's own stream (after it, nothing overlaps), 'n' = no expanders at all, 'd' = the expanders
// BEFORE the walkers (they time out waiting): in the last two every chunk marked for the expanders is given back to dec_expand_kernel
static char g_trail_debug = env_char("PCO_GFX_TRAIL_DEBUG");
static bool g_trail_always = env_first_is("PCO_GFX_DEC_TRAIL", '2');   // PCO_GFX_DEC_TRAIL=2: the expanders for calls of any size (tests)

// the workspace's second stream and the two events that fork it off the caller's stream and join it again
static void ensure_side_stream(Workspace& ws) {
  if (!ws.side_stream) PCO_HIP_CHECK(hipStreamCreateWithFlags(&ws.side_stream, hipStreamNonBlocking));
  if (!ws.side_stream2) PCO_HIP_CHECK(hipStreamCreateWithFlags(&ws.side_stream2, hipStreamNonBlocking));
  if (!ws.join_event2) PCO_HIP_CHECK(hipEventCreateWithFlags(&ws.join_event2, hipEventDisableTiming));
  if (!ws.fork_event) PCO_HIP_CHECK(hipEventCreateWithFlags(&ws.fork_event, hipEventDisableTiming));
  if (!ws.join_event) PCO_HIP_CHECK(hipEventCreateWithFlags(&ws.join_event, hipEventDisableTiming));
  if (!ws.n_cus) { hipDeviceProp_t prop; PCO_HIP_CHECK(hipGetDeviceProperties(&prop, ws.device)); ws.n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256; }
}

// metas (HOST array of n_tasks entries, or nullptr): where each PCO_GFX_TASK_WRAPPED_PAGE task's ChunkMeta lives when it is not in front of the page
static void launch_decode(size_t n_tasks, const PcoGfxDecodeTask* tasks, PcoGfxTaskResult* results,
                          PcoGfxTaskResult* d_results_user, hipStream_t stream, const MetaRef* metas = nullptr) {
  if (n_tasks == 0) return;
  Workspace& ws = workspace();
  // group task ids by number width: one kernel instantiation per width present in the batch
  std::vector<uint32_t> ids[4];
  bool any_bad = false;
  for (size_t i = 0; i < n_tasks; i++) {
    const int b = dtype_bits(tasks[i].dtype);
    if (b == 64) ids[0].push_back((uint32_t)i); else if (b == 32) ids[1].push_back((uint32_t)i); else if (b == 16) ids[2].push_back((uint32_t)i); else if (b == 8) ids[3].push_back((uint32_t)i); else any_bad = true;
  }
  if (any_bad) throw HostError{PCO_GFX_INVALID_ARGUMENT, "decode: invalid number type"};
  const size_t task_bytes = n_tasks * sizeof(PcoGfxDecodeTask);
  const size_t ids_off = (task_bytes + 15) & ~(size_t)15, metas_off = (ids_off + n_tasks * sizeof(uint32_t) + 15) & ~(size_t)15;
  uint8_t* d_base = (uint8_t*)ws.tasks.ensure(metas_off + (metas ? n_tasks * sizeof(MetaRef) : 0) + 64);
  PcoGfxDecodeTask* d_tasks = (PcoGfxDecodeTask*)d_base;
  uint32_t* d_ids = (uint32_t*)(d_base + ids_off);
  const MetaRef* d_metas = nullptr;
  if (metas) { PCO_HIP_CHECK(hipMemcpyAsync(d_base + metas_off, metas, n_tasks * sizeof(MetaRef), hipMemcpyHostToDevice, stream)); d_metas = (const MetaRef*)(d_base + metas_off); }
  PcoGfxTaskResult* d_results = d_results_user ? d_results_user : (PcoGfxTaskResult*)ws.results.ensure(n_tasks * sizeof(PcoGfxTaskResult));
  PCO_HIP_CHECK(hipMemcpyAsync(d_tasks, tasks, task_bytes, hipMemcpyHostToDevice, stream));
  const bool mixed = (ids[0].size() != n_tasks) && (ids[1].size() != n_tasks) && (ids[2].size() != n_tasks) && (ids[3].size() != n_tasks);
  std::vector<uint32_t> flat;
  size_t id_off[4] = {0, 0, 0, 0};
  if (mixed) {
    for (int g = 0; g < 4; g++) { id_off[g] = flat.size(); flat.insert(flat.end(), ids[g].begin(), ids[g].end()); }
    PCO_HIP_CHECK(hipMemcpyAsync(d_ids, flat.data(), flat.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
  }
  const uint32_t budget = g_decode_lds_bytes - kLdsFixed;
  size_t max_grid = 0;
  // blocks of the general one-wave-per-task kernel: each owns kTblWsBytes (832 KB) of global table scratch for tANS tables beyond its LDS,
  // so the grid is what that scratch scales with -- 4096 blocks are four rounds of what the CUs hold at once, the kernel strides over the rest
  constexpr size_t kGeneralGrid = 4096;
  for (int g = 0; g < 4; g++) max_grid = std::max(max_grid, std::min<size_t>(ids[g].size(), kGeneralGrid));
  // (synchronous calls: the table scratch only in the second pass, for the tasks that asked for it -- kStatusNeedHist covers both kinds of scratch)
  uint8_t* tbl = results ? nullptr : (uint8_t*)ws.tbl_ws.ensure(max_grid * kTblWsBytes);
  // Fast path (decode_fast.hip): walk 8 chunks per wave, then expand one chunk per wave; whatever it cannot take
  // (multi-chunk streams, big tANS tables, wrapped pages, ...) is finished by the single-kernel decoder.
  uint64_t max_cap = 0; bool plain = true;
  for (size_t i = 0; i < n_tasks; i++) { max_cap = std::max<uint64_t>(max_cap, tasks[i].dst_cap); if (tasks[i].flags & PCO_GFX_TASK_META_ONLY) plain = false; }   // (wrapped pages walk and expand like chunks)
  const uint64_t sym_stride = ((max_cap + 255) & ~(uint64_t)255) + 256, offpos_stride = sym_stride / 256 + 2;
  const bool fast = g_decode_fast && plain && n_tasks * 3 * sym_stride <= ((size_t)48 << 30);
  DecPlan* d_plans = nullptr; uint8_t* d_bins = nullptr; uint8_t* d_sym = nullptr; uint64_t* d_offpos = nullptr;
  if (fast) {
    d_plans = (DecPlan*)ws.dec_plans.ensure(n_tasks * sizeof(DecPlan));
    d_bins = (uint8_t*)ws.dec_bins.ensure(n_tasks * kBinsAreaPerTask);
    d_sym = (uint8_t*)ws.dec_sym.ensure(n_tasks * 3 * sym_stride + 64);
    d_offpos = (uint64_t*)ws.dec_offpos.ensure(n_tasks * 3 * offpos_stride * 8);
  }
  const uint32_t need_hist = results ? kStatusNeedHist : (uint32_t)PCO_GFX_UNSUPPORTED;   // (only a synchronous call can come back with the scratch)
  for (int g = 0; g < 4; g++) {
    if (ids[g].empty()) continue;
    const uint32_t cnt = (uint32_t)ids[g].size();
    const uint32_t grid = (uint32_t)std::min<size_t>(cnt, kGeneralGrid);
    const uint32_t* idp = mixed ? d_ids + id_off[g] : nullptr;
    const uint32_t* filt = fast ? (const uint32_t*)d_plans : nullptr;
    const uint32_t fstride = (uint32_t)(sizeof(DecPlan) / 4);
    if (fast) {  // walk with 8 chunks per wave (the common chunks expanded on the side stream meanwhile), then with 4 for the chunks whose tables did not fit, then expand the rest
      const uint32_t n_wb = (cnt + 7) / 8;
      uint32_t* d_progress = nullptr;
      uint32_t* d_givebacks = nullptr;   // device counter: chunks marked for the expanders under the walk that dec_expand_kernel had to take (pco_gfx_trail_givebacks)
      // The expanders under the walk pay from about a thousand chunks on: the publishing walker's chain is ~0.9 ms longer than the plain
      // one's whatever the call holds, and that buys the expansion -- 0.7 us a chunk at scale.  Smaller calls (the host-buffer entry points
      // decode one chunk per call) walk, then expand.
      constexpr uint32_t kTrailMinChunks = 1024;
      const bool use_trail = g_decode_trail && (cnt >= kTrailMinChunks || g_trail_debug != '\0' || g_trail_always);
      if (use_trail) {
        ensure_side_stream(ws);
        d_progress = (uint32_t*)ws.dec_progress.ensure((size_t)n_wb * kTrailProgressStride * sizeof(uint32_t));
        PCO_HIP_CHECK(hipMemsetAsync(d_progress, 0, (size_t)n_wb * kTrailProgressStride * sizeof(uint32_t), stream));
        if (!ws.dec_stats.p) { ws.dec_stats.ensure(256); PCO_HIP_CHECK(hipMemsetAsync(ws.dec_stats.p, 0, 256, stream)); }   // (in stream order: nothing synchronous inside an asynchronous call)
        d_givebacks = (uint32_t*)ws.dec_stats.p;
      }
      // (persistent expander grid: at most four blocks of four waves per CU, so that every walker block finds its wave slot, registers and LDS
      //  whatever the order in which the two kernels' blocks arrive)
      const uint32_t trail_grid = use_trail ? std::min<uint32_t>(n_wb, (uint32_t)ws.n_cus * 4u) : 0u;
#define PCO_FAST_DECODE(L, name)                                                                                                                          \
      {                                                                                                                                                   \
        static const bool _walk_ok = hipFuncSetAttribute((const void*)dec_walk_kernel<L, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * WalkCfg<8>::kWalkLdsBytes)) == hipSuccess && \
                                     hipFuncSetAttribute((const void*)dec_walk_kernel<L, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * WalkCfg<4>::kWalkLdsBytes)) == hipSuccess;   \
        if (!_walk_ok) throw HostError{PCO_GFX_DEVICE_ERROR, "cannot reserve LDS for dec_walk_kernel"};                                                   \
      }                                                                                                                                                   \
      if (use_trail) {                                                                                                                                    \
        ScopedKernelTimer _span("dec_walk+trail<" name ">", stream);                                                                                      \
        if (g_trail_debug == 'd') {   /* test switch: the expanders BEFORE the walkers on the caller's stream -- every wave times out waiting for a walker that has not started */ \
          hipLaunchKernelGGL((dec_trail_kernel<L, false>), dim3(trail_grid), dim3(64 * kTrailWaves), 0, stream, d_tasks, idp, cnt, d_plans, d_bins, d_sym, sym_stride, d_offpos, offpos_stride, d_progress, n_wb, d_metas); \
          hipLaunchKernelGGL((dec_trail_kernel<L, true>), dim3(trail_grid), dim3(64 * kTrailWaves), 0, stream, d_tasks, idp, cnt, d_plans, d_bins, d_sym, sym_stride, d_offpos, offpos_stride, d_progress, n_wb, d_metas);  \
        }                                                                                                                                                 \
        PCO_HIP_CHECK(hipEventRecord(ws.fork_event, stream));                                                                                             \
        /* four walker waves per workgroup, the CU's whole LDS: one walker per SIMD by construction (decode_fast.hip, walk_lds) */                      \
        static const bool _quad_ok = hipFuncSetAttribute((const void*)dec_walk_trail_kernel<L>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * WalkCfg<8>::kWalkLdsBytes)) == hipSuccess; \
        if (!_quad_ok) throw HostError{PCO_GFX_DEVICE_ERROR, "cannot reserve LDS for dec_walk_trail_kernel"};                                             \
        PCO_TIMED_LAUNCH("~dec_walk_kernel<" name ">", stream, (dec_walk_trail_kernel<L>), dim3((n_wb + 3) / 4), dim3(256), 4 * WalkCfg<8>::kWalkLdsBytes, stream, \
                         d_tasks, idp, cnt, d_plans, d_bins, d_sym, sym_stride, d_offpos, offpos_stride, d_results, d_progress, d_metas);                 \
        /* the blocks without a candidate for the expanders: the ordinary walker, beside the two (every block runs in exactly one of the walkers).       \
           Launched second: where no block is a candidate the publishing walker's blocks say so and leave at once, and the expanders with them         \
           (launched first it held the LDS, they queued behind it: 1.8 instead of 1.3 ms per 16384 one-bin chunks).  Where every block is a        \
           candidate its own blocks queue behind the publishing walker's LDS and then leave at once: its time in a profile is that wait */          \
        PCO_HIP_CHECK(hipStreamWaitEvent(ws.side_stream2, ws.fork_event, 0));                                                                             \
        PCO_TIMED_LAUNCH("~dec_walk_kernel(rest)<" name ">", ws.side_stream2, (dec_walk_kernel<L, 8>), dim3((n_wb + 3) / 4), dim3(256), 4 * WalkCfg<8>::kWalkLdsBytes, ws.side_stream2, \
                         d_tasks, idp, cnt, d_plans, d_bins, d_sym, sym_stride, d_offpos, offpos_stride, 0u, d_results, d_progress, d_metas);             \
        PCO_HIP_CHECK(hipEventRecord(ws.join_event2, ws.side_stream2));                                                                                   \
        hipStream_t ts = g_trail_debug == 's' ? stream : ws.side_stream;                                                                                  \
        PCO_HIP_CHECK(hipStreamWaitEvent(ws.side_stream, ws.fork_event, 0));                                                                              \
        /* one kernel per kind of walker block (classic chunks only / a chunk with two latent variables among them), one after the other on the    \
           expanders' stream: the blocks of the kind a call does not have leave at once */                                                         \
        if (g_trail_debug != 'n' && g_trail_debug != 'd') PCO_TIMED_LAUNCH("~dec_trail_kernel<" name ">", ts, (dec_trail_kernel<L, false>), dim3(trail_grid), dim3(64 * kTrailWaves), 0, ts, \
                         d_tasks, idp, cnt, d_plans, d_bins, d_sym, sym_stride, d_offpos, offpos_stride, d_progress, n_wb, d_metas);                               \
        if (g_trail_debug != 'n' && g_trail_debug != 'd') PCO_TIMED_LAUNCH("~dec_trail2_kernel<" name ">", ts, (dec_trail_kernel<L, true>), dim3(trail_grid), dim3(64 * kTrailWaves), 0, ts, \
                         d_tasks, idp, cnt, d_plans, d_bins, d_sym, sym_stride, d_offpos, offpos_stride, d_progress, n_wb, d_metas);                               \
        PCO_HIP_CHECK(hipEventRecord(ws.join_event, ws.side_stream));                                                                                     \
        PCO_HIP_CHECK(hipStreamWaitEvent(stream, ws.join_event, 0));                                                                                      \
        PCO_HIP_CHECK(hipStreamWaitEvent(stream, ws.join_event2, 0));                                                                                     \
      } else PCO_TIMED_LAUNCH("dec_walk_kernel<" name ">", stream, (dec_walk_kernel<L, 8>), dim3((n_wb + 3) / 4), dim3(256), 4 * WalkCfg<8>::kWalkLdsBytes, stream, \
                       d_tasks, idp, cnt, d_plans, d_bins, d_sym, sym_stride, d_offpos, offpos_stride, 0u, d_results, (uint32_t*)nullptr, d_metas);       \
      PCO_TIMED_LAUNCH("dec_walk4_kernel<" name ">", stream, (dec_walk_kernel<L, 4>), dim3((cnt + 15) / 16), dim3(256), 4 * WalkCfg<4>::kWalkLdsBytes, stream, \
                       d_tasks, idp, cnt, d_plans, d_bins, d_sym, sym_stride, d_offpos, offpos_stride, kStatusRetryK4, d_results, (uint32_t*)nullptr, d_metas); \
      PCO_TIMED_LAUNCH("dec_expand_kernel<" name ">", stream, (dec_expand_kernel<L, false>), dim3(grid), dim3(256), kExpLdsBytes, stream,               \
                       d_tasks, d_results, idp, cnt, d_plans, d_bins, d_sym, sym_stride, d_offpos, offpos_stride, (const uint32_t*)d_progress, d_givebacks);    \
      PCO_TIMED_LAUNCH("dec_expand_lb_kernel<" name ">", stream, (dec_expand_kernel<L, true>), dim3(grid), dim3(256), kExpLbLdsBytes, stream,           \
                       d_tasks, d_results, idp, cnt, d_plans, d_bins, d_sym, sym_stride, d_offpos, offpos_stride, (const uint32_t*)d_progress, d_givebacks);
      if (g == 0) { PCO_FAST_DECODE(uint64_t, "u64") } else if (g == 1) { PCO_FAST_DECODE(uint32_t, "u32") } else if (g == 2) { PCO_FAST_DECODE(uint16_t, "u16") } else { PCO_FAST_DECODE(uint8_t, "u8") }
#undef PCO_FAST_DECODE
    }
    if (g == 0) PCO_TIMED_LAUNCH("pco_decode_kernel<u64>", stream, pco_decode_kernel<uint64_t>, dim3(grid), dim3(64), g_decode_lds_bytes, stream, d_tasks, d_results, idp, cnt, budget, tbl, filt, fstride, kStatusRetryLegacy, (uint8_t*)nullptr, (const uint64_t*)nullptr, need_hist, d_metas);
    else if (g == 1) PCO_TIMED_LAUNCH("pco_decode_kernel<u32>", stream, pco_decode_kernel<uint32_t>, dim3(grid), dim3(64), g_decode_lds_bytes, stream, d_tasks, d_results, idp, cnt, budget, tbl, filt, fstride, kStatusRetryLegacy, (uint8_t*)nullptr, (const uint64_t*)nullptr, need_hist, d_metas);
    else if (g == 2) PCO_TIMED_LAUNCH("pco_decode_kernel<u16>", stream, pco_decode_kernel<uint16_t>, dim3(grid), dim3(64), g_decode_lds_bytes, stream, d_tasks, d_results, idp, cnt, budget, tbl, filt, fstride, kStatusRetryLegacy, (uint8_t*)nullptr, (const uint64_t*)nullptr, need_hist, d_metas);
    else PCO_TIMED_LAUNCH("pco_decode_kernel<u8>", stream, pco_decode_kernel<uint8_t>, dim3(grid), dim3(64), g_decode_lds_bytes, stream, d_tasks, d_results, idp, cnt, budget, tbl, filt, fstride, kStatusRetryLegacy, (uint8_t*)nullptr, (const uint64_t*)nullptr, need_hist, d_metas);
    PCO_HIP_CHECK(hipGetLastError());
  }
  if (results) {
    PCO_HIP_CHECK(hipMemcpyAsync(results, d_results, n_tasks * sizeof(PcoGfxTaskResult), hipMemcpyDeviceToHost, stream));
    PCO_HIP_CHECK(hipStreamSynchronize(stream));
    // Tasks handed back for want of scratch (tANS tables beyond the LDS budget; a lookback delta whose secondary variable is delta'd too: its history needs dst_cap latents):
    // once more through the single-kernel decoder, with the scratch.  The format allows the combination; no encoder writes it.
    std::vector<uint32_t> again[4]; size_t n_again = 0;
    for (size_t i = 0; i < n_tasks; i++) if (results[i].status == kStatusNeedHist) {
      const int b = dtype_bits(tasks[i].dtype);
      again[b == 64 ? 0 : (b == 32 ? 1 : (b == 16 ? 2 : 3))].push_back((uint32_t)i); n_again++;
    }
    if (n_again) {
      std::vector<uint32_t> flat_ids; std::vector<uint64_t> offs; uint64_t hist_bytes = 0;
      size_t goff[4];
      for (int g = 0; g < 4; g++) { goff[g] = flat_ids.size(); for (uint32_t i : again[g]) { flat_ids.push_back(i); offs.push_back(hist_bytes); hist_bytes += ((tasks[i].dst_cap * (uint64_t)(dtype_bits(tasks[i].dtype) / 8)) + 255) & ~(uint64_t)255; } }
      uint8_t* d_hist = (uint8_t*)ws.dec_hist.ensure(hist_bytes + n_again * 12 + 256);
      uint64_t* d_hoff = (uint64_t*)(d_hist + ((hist_bytes + 15) & ~(uint64_t)15));
      uint32_t* d_again = (uint32_t*)(d_hoff + n_again);
      PCO_HIP_CHECK(hipMemcpyAsync(d_hoff, offs.data(), n_again * 8, hipMemcpyHostToDevice, stream));
      PCO_HIP_CHECK(hipMemcpyAsync(d_again, flat_ids.data(), n_again * 4, hipMemcpyHostToDevice, stream));
      uint8_t* tbl2 = (uint8_t*)ws.tbl_ws.ensure(std::min<size_t>(n_again, kGeneralGrid) * kTblWsBytes);
      for (int g = 0; g < 4; g++) {
        if (again[g].empty()) continue;
        const uint32_t cnt = (uint32_t)again[g].size(), grid = (uint32_t)std::min<size_t>(cnt, kGeneralGrid);
        const uint32_t* idp = d_again + goff[g]; const uint64_t* hop = d_hoff + goff[g];
        if (g == 0) PCO_TIMED_LAUNCH("pco_decode_kernel<u64>", stream, pco_decode_kernel<uint64_t>, dim3(grid), dim3(64), g_decode_lds_bytes, stream, d_tasks, d_results, idp, cnt, budget, tbl2, (const uint32_t*)nullptr, 0u, 0u, d_hist, hop, (uint32_t)PCO_GFX_UNSUPPORTED, d_metas);
        else if (g == 1) PCO_TIMED_LAUNCH("pco_decode_kernel<u32>", stream, pco_decode_kernel<uint32_t>, dim3(grid), dim3(64), g_decode_lds_bytes, stream, d_tasks, d_results, idp, cnt, budget, tbl2, (const uint32_t*)nullptr, 0u, 0u, d_hist, hop, (uint32_t)PCO_GFX_UNSUPPORTED, d_metas);
        else if (g == 2) PCO_TIMED_LAUNCH("pco_decode_kernel<u16>", stream, pco_decode_kernel<uint16_t>, dim3(grid), dim3(64), g_decode_lds_bytes, stream, d_tasks, d_results, idp, cnt, budget, tbl2, (const uint32_t*)nullptr, 0u, 0u, d_hist, hop, (uint32_t)PCO_GFX_UNSUPPORTED, d_metas);
        else PCO_TIMED_LAUNCH("pco_decode_kernel<u8>", stream, pco_decode_kernel<uint8_t>, dim3(grid), dim3(64), g_decode_lds_bytes, stream, d_tasks, d_results, idp, cnt, budget, tbl2, (const uint32_t*)nullptr, 0u, 0u, d_hist, hop, (uint32_t)PCO_GFX_UNSUPPORTED, d_metas);
      }
      PCO_HIP_CHECK(hipGetLastError());
      PCO_HIP_CHECK(hipMemcpyAsync(results, d_results, n_tasks * sizeof(PcoGfxTaskResult), hipMemcpyDeviceToHost, stream));
      PCO_HIP_CHECK(hipStreamSynchronize(stream));
    }
  }
}

}  // namespace pcogfx

using namespace pcogfx;

extern "C" {

int pco_gfx_last_status(void) { return g_err.status; }
const char* pco_gfx_last_error(void) { return g_err.msg.c_str(); }
int pco_gfx_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }
void pco_gfx_release_workspace(void) {   // this thread's workspace on the current device
  try { Workspace& w = workspace(); if (w.has_last && w.last_event) (void)hipEventSynchronize(w.last_event); w.release_all(); } catch (...) {}
}

size_t pco_gfx_workspace_bytes(void) {
  try {
    static const bool trace = std::getenv("PCO_GFX_TRACE") != nullptr;
    if (trace) fprintf(stderr, "[pco_gfx trace] workspace: %s\n", workspace().device_report().c_str());
    return workspace().device_bytes();
  } catch (...) { return 0; }
}

static unsigned long long trail_stat(int which) {
  try {
    Workspace& w = workspace();
    const unsigned long long acc = which == 0 ? w.acc_givebacks : w.acc_marked;
    if (!w.dec_stats.p) return acc;
    if (w.has_last && w.last_event) (void)hipEventSynchronize(w.last_event);
    uint32_t v[2] = {0, 0};
    if (hipMemcpy(v, w.dec_stats.p, 8, hipMemcpyDeviceToHost) != hipSuccess) return acc;
    return acc + v[which];
  } catch (...) { return 0; }
}
unsigned long long pco_gfx_trail_givebacks(void) { return trail_stat(0); }
unsigned long long pco_gfx_trail_marked(void) { return trail_stat(1); }
unsigned long long pco_gfx_strict_histogram_fallbacks(void) {
  try {
    Workspace& w = workspace();
    if (!w.enc_strict.p) return w.acc_strict;
    if (w.has_last && w.last_event) (void)hipEventSynchronize(w.last_event);
    uint32_t v = 0;
    if (hipMemcpy(&v, w.enc_strict.p, 4, hipMemcpyDeviceToHost) != hipSuccess) return w.acc_strict;
    return w.acc_strict + v;
  } catch (...) { return 0; }
}

// Kernel timing: begin() arms per-launch HIP events on this thread; end() synchronises and returns one
// (name, milliseconds) pair per kernel launched since begin().  Names are written NUL-separated.
void pco_gfx_profile_begin(void) {
  for (auto& r : g_prof.recs) { g_prof.give(r.a); g_prof.give(r.b); }
  g_prof.recs.clear(); g_prof.on = true;
}
int pco_gfx_profile_end(char* names, size_t names_cap, float* ms, int cap) {
  g_prof.on = false;
  int n = 0; size_t pos = 0;
  for (auto& r : g_prof.recs) {
    float t = 0.f;
    (void)hipEventSynchronize(r.b);
    (void)hipEventElapsedTime(&t, r.a, r.b);
    if (n < cap) {
      const size_t len = std::strlen(r.name) + 1;
      if (names && pos + len <= names_cap) { std::memcpy(names + pos, r.name, len); pos += len; }
      if (ms) ms[n] = t;
      n++;
    }
    g_prof.give(r.a); g_prof.give(r.b);
  }
  g_prof.recs.clear();
  return n;
}

size_t pco_gfx_guarantee_file_size(size_t n, unsigned char dtype, uint64_t max_page_n) {
  const int bits = dtype_bits(dtype);
  if (!bits) return 0;
  std::vector<size_t> pages;
  n_per_page(max_page_n, n, pages);
  size_t res = guarantee_standalone_header_size() + 1;
  for (size_t p : pages) res += guarantee_standalone_chunk_size(bits, p);
  return res;
}
size_t pco_standalone_guarantee_file_size(size_t n, unsigned char dtype) { return pco_gfx_guarantee_file_size(n, dtype, 0); }
size_t pco_gfx_guarantee_chunk_size(size_t n, unsigned char dtype) {
  const int bits = dtype_bits(dtype);
  return bits ? guarantee_standalone_chunk_size(bits, n) : 0;
}

size_t pco_gfx_write_standalone_header(void* dst, size_t dst_cap, uint64_t n_hint, unsigned char uniform_dtype) {
  // standalone/compressor.rs:12-16,85-105 + wrapped/file_compressor.rs:54-59
  HostBitWriter w;
  w.write(0x216f6370u, 32); w.write(3, 8); w.write(uniform_dtype, 8);
  const uint32_t power = n_hint == 0 ? 1 : (64 - (uint32_t)__builtin_clzll(n_hint));
  w.write(power - 1, kBitsVarintPower); w.write(n_hint, power); w.finish_byte();
  w.write(4, 8); w.write(1, 8);
  if (w.bytes() > dst_cap) return 0;
  std::memcpy(dst, w.buf.data(), w.bytes());
  return w.bytes();
}
size_t pco_gfx_write_standalone_footer(void* dst, size_t dst_cap) {
  if (dst_cap < 1) return 0;
  *(uint8_t*)dst = 0;
  return 1;
}
size_t pco_wrapped_write_header(void* dst, size_t dst_cap) {
  if (dst_cap < 2) return 0;
  ((uint8_t*)dst)[0] = 4; ((uint8_t*)dst)[1] = 1;
  return 2;
}
enum PcoError pco_wrapped_read_header(const void* src, size_t len, size_t* consumed, uint8_t* major, uint8_t* minor) {
  clear_error();
  const uint8_t* p = (const uint8_t*)src;
  if (len < 1) { set_error(PCO_GFX_INSUFFICIENT_DATA, "empty header"); return PcoDecompressionError; }
  uint8_t mj = p[0], mn = 0; size_t used = 1;
  if (mj >= 4) { if (len < 2) { set_error(PCO_GFX_INSUFFICIENT_DATA, "short header"); return PcoDecompressionError; } mn = p[1]; used = 2; }
  if (mj > 4) { set_error(PCO_GFX_CORRUPTION, "file's format version definitely cannot be decompressed"); return PcoDecompressionError; }
  if (consumed) *consumed = used; if (major) *major = mj; if (minor) *minor = mn;
  return PcoSuccess;
}

enum PcoError pco_gfx_decompress_chunks(size_t n_tasks, const PcoGfxDecodeTask* tasks, PcoGfxTaskResult* results,
                                        PcoGfxTaskResult* d_results, void* stream) {
  clear_error();
  try {
    require_device();
    { WorkspaceUse use(workspace(), (hipStream_t)stream); launch_decode(n_tasks, tasks, results, d_results, (hipStream_t)stream); }
    if (results) for (size_t i = 0; i < n_tasks; i++) if (results[i].status != PCO_GFX_OK) {
      set_error((int)results[i].status, "decode task " + std::to_string(i) + " failed");
      return PcoDecompressionError;
    }
    return PcoSuccess;
  } catch (const HostError& e) { return fail_with(e, PcoDecompressionError); }
}

enum PcoError pco_gfx_decompress_pages(size_t n_tasks, const PcoGfxPageTask* tasks, PcoGfxTaskResult* results, PcoGfxTaskResult* d_results, void* stream) {
  clear_error();
  try {
    require_device();
    std::vector<PcoGfxDecodeTask> dt(n_tasks); std::vector<MetaRef> refs(n_tasks);
    for (size_t i = 0; i < n_tasks; i++) {
      const PcoGfxPageTask& t = tasks[i];
      if (t.format_major > 4) throw HostError{PCO_GFX_CORRUPTION, "page task " + std::to_string(i) + ": the file's format version definitely cannot be decompressed"};   // wrapped/file_decompressor.rs:31-36
      if (t.meta == nullptr || t.page == nullptr) throw HostError{PCO_GFX_INVALID_ARGUMENT, "page task " + std::to_string(i) + ": null ChunkMeta or page"};
      dt[i] = PcoGfxDecodeTask{t.page, t.page_len, t.dst, t.page_n, t.dtype, PCO_GFX_TASK_WRAPPED_PAGE | (t.format_major << 8)};
      refs[i] = MetaRef{t.meta, t.meta_len};
    }
    { WorkspaceUse use(workspace(), (hipStream_t)stream); launch_decode(n_tasks, dt.data(), results, d_results, (hipStream_t)stream, refs.data()); }
    if (results) for (size_t i = 0; i < n_tasks; i++) if (results[i].status != PCO_GFX_OK) {
      set_error((int)results[i].status, "page task " + std::to_string(i) + " failed");
      return PcoDecompressionError;
    }
    return PcoSuccess;
  } catch (const HostError& e) { return fail_with(e, PcoDecompressionError); }
}

enum PcoError pco_gfx_compact_chunks(size_t n_tasks, const PcoGfxEncodeTask* tasks, const PcoGfxTaskResult* d_results, void* d_dst,
                                     uint64_t dst_cap, uint64_t dst_offset, uint64_t* d_offsets, uint64_t* total, void* stream_) {
  clear_error();
  try {
    require_device();
    hipStream_t stream = (hipStream_t)stream_;
    if (!d_results || !d_offsets || (!d_dst && dst_cap)) throw HostError{PCO_GFX_INVALID_ARGUMENT, "compact: null device array"};
    if (n_tasks >= (1ull << 31)) throw HostError{PCO_GFX_INVALID_ARGUMENT, "compact: too many chunks"};
    Workspace& ws = workspace();
    WorkspaceUse use(ws, stream);
    uint8_t* d_base = (uint8_t*)ws.compact_tasks.ensure(n_tasks * sizeof(PcoGfxEncodeTask) + 64);
    uint32_t* d_over = (uint32_t*)d_base; PcoGfxEncodeTask* d_tasks = (PcoGfxEncodeTask*)(d_base + 64);
    uint64_t max_cap = 0;
    for (size_t i = 0; i < n_tasks; i++) max_cap = std::max<uint64_t>(max_cap, tasks[i].dst_cap);
    if (n_tasks) PCO_HIP_CHECK(hipMemcpyAsync(d_tasks, tasks, n_tasks * sizeof(PcoGfxEncodeTask), hipMemcpyHostToDevice, stream));
    PCO_TIMED_LAUNCH("compact_scan_kernel", stream, compact_scan_kernel, dim3(1), dim3(1024), 0, stream, d_results, (uint32_t)n_tasks, dst_offset, dst_cap, d_offsets, d_over);
    if (n_tasks) {
      const uint64_t slice = 64 * 1024;   // one block per 64 KiB of a chunk: >> 256 blocks in flight for any many-chunk call
      const uint32_t slices = (uint32_t)std::max<uint64_t>(1, (max_cap + slice - 1) / slice);
      if ((uint64_t)slices * n_tasks >= (1ull << 31)) throw HostError{PCO_GFX_INVALID_ARGUMENT, "compact: too many chunks for one call"};
      PCO_TIMED_LAUNCH("compact_copy_kernel", stream, compact_copy_kernel, dim3((uint32_t)(slices * n_tasks)), dim3(256), 0, stream, d_tasks, d_results, d_offsets, (uint8_t*)d_dst, d_over, (uint32_t)n_tasks, slice, slices);
    }
    PCO_HIP_CHECK(hipGetLastError());
    if (total) {
      uint32_t over = 0;
      PCO_HIP_CHECK(hipMemcpyAsync(total, d_offsets + n_tasks, 8, hipMemcpyDeviceToHost, stream));
      PCO_HIP_CHECK(hipMemcpyAsync(&over, d_over, 4, hipMemcpyDeviceToHost, stream));
      PCO_HIP_CHECK(hipStreamSynchronize(stream));
      if (over) throw HostError{PCO_GFX_INVALID_ARGUMENT, "compact: destination too small"};
    }
    return PcoSuccess;
  } catch (const HostError& e) { return fail_with(e, PcoCompressionError); }
}

// standalone/decompressor.rs:85-137 on the host: where the first chunk starts and which format version the file declares.  Anything out
// of the ordinary (old layouts without a version byte, a foreign uniform type, padding that is not zero, truncation) returns false and the
// device's own parser reports it.
static bool parse_standalone_header_host(const uint8_t* p, size_t len, unsigned char dtype, size_t& off, uint32_t& fmt_major) {
  if (len < 6 || p[0] != 'p' || p[1] != 'c' || p[2] != 'o' || p[3] != '!') return false;
  const uint32_t sv = p[4];
  if (sv < 2 || sv > 3) return false;
  size_t pos = 5;
  if (sv >= 3) { const uint32_t uniform = p[pos++]; if (uniform != 0 && uniform != dtype) return false; }
  if (pos + 9 > len) return false;
  uint64_t bits = 0; for (int i = 0; i < 9 && i < 8; i++) bits |= (uint64_t)p[pos + i] << (8 * i);
  const uint32_t power = 1 + (uint32_t)(bits & 63u);
  const uint32_t vbits = kBitsVarintPower + power, vbytes = (vbits + 7) / 8;
  if (vbits & 7) {   // the padding up to the byte boundary must be zero
    const uint32_t last = p[pos + vbytes - 1];
    if ((last >> (vbits & 7)) != 0) return false;
  }
  pos += vbytes;
  if (pos >= len) return false;
  fmt_major = p[pos++];
  if (fmt_major > 4) return false;
  if (fmt_major >= 4) pos++;
  if (pos > len) return false;
  off = pos;
  return true;
}

enum PcoError pco_standalone_simple_decompress_into(const void* compressed, size_t compressed_len, unsigned char dtype,
                                                    void* dst, size_t dst_cap, size_t* n_written) {
  clear_error();
  const int bits = dtype_bits(dtype);
  if (!bits) { set_error(PCO_GFX_INVALID_ARGUMENT, "invalid dtype"); return PcoInvalidType; }
  try {
    require_device();
    Workspace& ws = workspace();
    WorkspaceUse use(ws, 0);   // (ordered behind an asynchronous batched call of this thread that may still be running on another stream)
    uint8_t* d_in = (uint8_t*)ws.io_in.ensure(compressed_len + 64);
    const size_t esz = (size_t)(bits / 8), out_bytes = dst_cap * esz;
    uint8_t* d_out = (uint8_t*)ws.io_out.ensure(out_bytes + 64);
    PCO_HIP_CHECK(hipMemsetAsync(d_in + compressed_len, 0, 64, 0));
    if (compressed_len) PCO_HIP_CHECK(hipMemcpyAsync(d_in, compressed, compressed_len, hipMemcpyHostToDevice, 0));
    // A file stores no chunk lengths: chunk k + 1 starts where chunk k's bit stream ends, so the chunks of one file decode one after the
    // other whatever does the decoding (standalone/decompressor.rs:233-301).  Each goes through the two-kernel path on its own (one
    // PCO_GFX_TASK_ONE_CHUNK task per chunk, ~4 ms of tANS chain latency each); handing the whole file to the single-kernel decoder, one
    // wave walking and expanding chunk after chunk, cost 14 ms per 2^18-number chunk.
    size_t off = 0, done = 0; uint32_t fmt_major = 4;
    const bool by_chunk = g_decode_fast && parse_standalone_header_host((const uint8_t*)compressed, compressed_len, dtype, off, fmt_major);
    if (!by_chunk) {
      PcoGfxDecodeTask task{d_in, compressed_len, d_out, dst_cap, dtype, PCO_GFX_TASK_HAS_FILE_HEADER};
      PcoGfxTaskResult res{};
      launch_decode(1, &task, &res, nullptr, 0);
      if (res.status != PCO_GFX_OK) {
        // a too-small dst is PcoDecompressionError, like pco_c/src/lib.rs:110-112
        set_error((int)res.status, "decompression failed");
        return PcoDecompressionError;
      }
      done = res.n_out;
    } else {
      for (;;) {
        if (off >= compressed_len) { set_error(PCO_GFX_INSUFFICIENT_DATA, "decompression failed: the file ends without its terminator"); return PcoDecompressionError; }
        if (((const uint8_t*)compressed)[off] == 0) break;   // the terminator where a chunk's type byte would be: a file without (further) chunks (standalone/decompressor.rs:190-200)
        // (a chunk holds at most 2^24 numbers: the scratch launch_decode sizes from dst_cap stays a chunk's, whatever the caller's buffer)
        PcoGfxDecodeTask task{d_in + off, compressed_len - off, d_out + done * esz, std::min<size_t>(dst_cap - done, kMaxEntries), dtype, PCO_GFX_TASK_ONE_CHUNK | (fmt_major << 8)};
        PcoGfxTaskResult res{};
        launch_decode(1, &task, &res, nullptr, 0);
        if (res.status != PCO_GFX_OK) { set_error((int)res.status, "decompression failed"); return PcoDecompressionError; }
        done += res.n_out; off += res.consumed;
        if (res.aux & 1u) { if (res.consumed == 0) break; continue; }   // another chunk follows
        // the last chunk: the file must still hold its terminator byte (standalone/decompressor.rs:190-200: chunk_preamble fails with
        // InsufficientData on a file that ends right behind a chunk); the kernels report it consumed through aux bit 1
        if (!(res.aux & 2u)) { set_error(PCO_GFX_INSUFFICIENT_DATA, "decompression failed: the file ends without its terminator"); return PcoDecompressionError; }
        break;
      }
    }
    if (done) PCO_HIP_CHECK(hipMemcpy(dst, d_out, done * esz, hipMemcpyDeviceToHost));
    if (n_written) *n_written = done;
    return PcoSuccess;
  } catch (const HostError& e) { return fail_with(e, PcoDecompressionError); }
}

}  // extern "C"

#include "pco_gfx_encode_api.inc"
#include "pco_gfx_comm.inc"

#ifdef PCO_TRAIL_TIMING
extern "C" int pco_gfx_debug_trail_timing(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pcogfx::g_trail_timing), 64);
}
extern "C" int pco_gfx_debug_trail_stamps(unsigned long long* out) {   // [4][kTrailStampBlocks]
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pcogfx::g_trail_stamps), sizeof(unsigned long long) * 4 * pcogfx::kTrailStampBlocks);
}
#endif
#ifdef PCO_WP_ASSERT
extern "C" int pco_gfx_debug_wp_err(uint32_t* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pcogfx::g_wp_err), 32); }
#endif
#ifdef PCO_WP_DEBUGSUM
extern "C" int pco_gfx_debug_wp_sums(uint32_t* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pcogfx::g_wp_dbg), 16 * 1100 * 16); }
#endif
#ifdef PCO_WP_TIMING
extern "C" int pco_gfx_debug_wp_timing(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pcogfx::g_wp_timing), 128); }
#endif
#ifdef PCO_WALK_TIMING
extern "C" int pco_gfx_debug_walk_timing(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pcogfx::g_walk_timing), 64);
}
#endif

// Test hook: stage 1 of Auto mode detection on floats (auto_float_stats_kernel) on a host array taken as the sample, in order.
// out: s_size, tz5, n_gcd, sim[3], hist[56], then has_euclid, k, n_ints, base_c (lo, hi).  Lets the GPU tests check the device's
// arithmetic against an IEEE reference (numpy).
extern "C" int pco_gfx_debug_float_screen(const void* values, size_t n, uint32_t dtype, uint32_t* out) {
  using namespace pcogfx;
  if (n == 0 || n > kAutoCap || dtype_kind(dtype) != kFloat || dtype_bits(dtype) < 32) return PCO_GFX_INVALID_ARGUMENT;
  const size_t eb = dtype_bits(dtype) / 8;
  void* d_vals = nullptr; void* d_sbuf = nullptr; uint32_t* d_idx = nullptr; FloatStatsTask* d_task = nullptr; FloatStatsResult* d_res = nullptr;
  std::vector<uint32_t> idx(n); for (size_t i = 0; i < n; i++) idx[i] = (uint32_t)i;
  int rc = PCO_GFX_OK;
  if (hipFuncSetAttribute((const void*)auto_float_stats_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kAutoStage1LdsBytes) != hipSuccess) return PCO_GFX_DEVICE_ERROR;
  if (hipMalloc(&d_vals, n * eb) != hipSuccess || hipMalloc(&d_sbuf, kAutoCap * 8) != hipSuccess || hipMalloc((void**)&d_idx, n * 4) != hipSuccess || hipMalloc((void**)&d_task, sizeof(FloatStatsTask)) != hipSuccess || hipMalloc((void**)&d_res, sizeof(FloatStatsResult)) != hipSuccess) rc = PCO_GFX_DEVICE_ERROR;
  if (rc == PCO_GFX_OK) {
    const FloatStatsTask t{d_vals, d_idx, d_sbuf, (uint32_t)n, dtype};
    static FloatStatsResult r;
    if (hipMemcpy(d_vals, values, n * eb, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_idx, idx.data(), n * 4, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_task, &t, sizeof(t), hipMemcpyHostToDevice) != hipSuccess) rc = PCO_GFX_DEVICE_ERROR;
    else {
      hipLaunchKernelGGL(auto_float_stats_kernel, dim3(1), dim3(kAutoT), kAutoStage1LdsBytes, 0, d_task, d_res);
      if (hipMemcpy(&r, d_res, sizeof(r), hipMemcpyDeviceToHost) != hipSuccess) rc = PCO_GFX_DEVICE_ERROR;
      else {
        out[0] = r.s_size; out[1] = r.tz5; out[2] = r.n_gcd; out[3] = r.sim[0]; out[4] = r.sim[1]; out[5] = r.sim[2]; for (int i = 0; i < 56; i++) out[6 + i] = r.hist[i];
        out[62] = r.has_euclid; out[63] = (uint32_t)r.k; out[64] = r.n_ints; out[65] = (uint32_t)r.base_c; out[66] = (uint32_t)(r.base_c >> 32);
      }
    }
  }
  (void)hipFree(d_vals); (void)hipFree(d_sbuf); (void)hipFree(d_idx); (void)hipFree(d_task); (void)hipFree(d_res);
  return rc;
}

#ifdef PCO_SEL_TIMING
extern "C" int pco_gfx_debug_sel_timing(unsigned long long* out, int reset) {
  if (reset) { unsigned long long z[16] = {}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(pcogfx::g_sel_timing), z, sizeof(z)); }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pcogfx::g_sel_timing), 128);
}
#endif

#ifdef PCO_OCCUPANCY_PROBE
namespace pcogfx {
__global__ void xor_lane_check_kernel(uint32_t* bad) {
  const uint32_t lane = threadIdx.x & 63u; const uint32_t v = 0x9e3779b9u * (threadIdx.x + 1u); const uint64_t w = ((uint64_t)v << 32) | (v ^ 0x5555u);
  uint32_t n = 0;
  n += xor_lane<1>(v) != (uint32_t)__shfl((int)v, (int)(lane ^ 1), 64); n += xor_lane<2>(v) != (uint32_t)__shfl((int)v, (int)(lane ^ 2), 64);
  n += xor_lane<4>(v) != (uint32_t)__shfl((int)v, (int)(lane ^ 4), 64); n += xor_lane<8>(v) != (uint32_t)__shfl((int)v, (int)(lane ^ 8), 64);
  n += xor_lane<16>(v) != (uint32_t)__shfl((int)v, (int)(lane ^ 16), 64); n += xor_lane<32>(v) != (uint32_t)__shfl((int)v, (int)(lane ^ 32), 64);
  n += xor_lane<16>(w) != shfl_idx(w, (int)(lane ^ 16)); n += xor_lane<4>(w) != shfl_idx(w, (int)(lane ^ 4));
  uint64_t key = (uint64_t)(v >> 7) * 2654435761ull; const uint64_t sorted = wave_sort64<uint64_t>(key);
  const uint64_t prev = shfl_idx(sorted, (int)(lane == 0 ? 0 : lane - 1)); n += sorted < prev;
  if (n) atomicAdd(bad, n);
}
}
extern "C" int pco_gfx_debug_xor_lane_check(unsigned* bad_out) {
  uint32_t* d = nullptr; if (hipMalloc(&d, 4) != hipSuccess) return -1;
  (void)hipMemset(d, 0, 4);
  hipLaunchKernelGGL(pcogfx::xor_lane_check_kernel, dim3(4), dim3(256), 0, 0, d);
  const int rc = (int)hipMemcpy(bad_out, d, 4, hipMemcpyDeviceToHost); (void)hipFree(d); return rc;
}
// resident blocks per CU of the kernels whose design counts on a number (scripts/occupancy.py, variant build only)
extern "C" int pco_gfx_debug_occupancy(int which, int* blocks) {
  using namespace pcogfx;
  hipError_t e = hipErrorInvalidValue;
  if (which == 0) { (void)hipFuncSetAttribute((const void*)enc_hist_select_kernel<uint32_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSelLdsBytes); e = hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, enc_hist_select_kernel<uint32_t>, (int)kSelThr, kSelLdsBytes); }
  else if (which == 1) { (void)hipFuncSetAttribute((const void*)enc_hist_wide_kernel<kMidHistRange>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hist_lds_bytes(kMidHistRange)); e = hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, enc_hist_wide_kernel<kMidHistRange>, 1024, hist_lds_bytes(kMidHistRange)); }
  else if (which == 2) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, enc_walk_kernel<8>, 64, EwCfg<8>::kLdsBytes);
  else if (which == 3) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, enc_pack_kernel, 64, 6144);
  return (int)e;
}
#endif

#ifdef PCO_WS_TRACE
extern "C" int pco_gfx_debug_ws_trace(unsigned long long* out, int n_blocks) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pcogfx::g_ws_trace), (size_t)n_blocks * 24); }
#endif

#ifdef PCO_HIST_TIMING
extern "C" int pco_gfx_debug_hist_timing(unsigned long long* out, int reset) {
  if (reset) { unsigned long long z[16] = {}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(pcogfx::g_hist_timing), z, sizeof(z)); }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pcogfx::g_hist_timing), 128);
}
#endif

#ifdef PCO_LBP_TIMING
extern "C" int pco_gfx_debug_lbp_timing(unsigned long long* out, int reset) {
  if (reset) { unsigned long long z[16] = {}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(pcogfx::g_lbp_timing), z, sizeof(z)); }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pcogfx::g_lbp_timing), 128);
}
#endif

#ifdef PCO_LB_TIMING
extern "C" int pco_gfx_debug_lb_timing(unsigned long long* out, int reset) {
  if (reset) { unsigned long long z[16] = {}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(pcogfx::g_lb_timing), z, sizeof(z)); }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pcogfx::g_lb_timing), 128);
}
#endif
