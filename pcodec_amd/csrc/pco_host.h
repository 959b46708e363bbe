// pco_host.h -- host-side plumbing of libpco_gfx.so: error state, the per-thread device
// workspace, and the host-side (tiny, O(metadata)) framing helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/pco_gfx.h"

namespace pcogfx {

struct HostError {
  int status;
  std::string msg;
  bool oom = false;   // a device / pinned allocation failed: the sub-batching encoder retries with fewer chunks per pass
};

void set_error(int status, const std::string& msg);
void clear_error();

// Environment switches (A/B runs, tests) are read through these NAMED functions, never through a lambda in an initialiser: hipcc numbers the
// lambdas of the host and device passes of this translation unit differently, and a namespace-scope `static const bool x = [] {...}();` was
// once compiled with ANOTHER lambda's body (round 5: every call silently ran strict histograms).  tests/test_gpu_switch_defaults.py asserts
// every switch's default with a clean environment.
inline bool env_is_one(const char* name) { const char* e = std::getenv(name); return e != nullptr && e[0] == '1'; }
inline bool env_not_zero(const char* name) { const char* e = std::getenv(name); return !(e != nullptr && e[0] == '0'); }   // default ON, "0..." switches off
inline char env_char(const char* name) { const char* e = std::getenv(name); return e != nullptr ? e[0] : '\0'; }
inline bool env_first_is(const char* name, char c) { const char* e = std::getenv(name); return e != nullptr && e[0] == c; }
inline int env_int(const char* name, int dflt) { const char* e = std::getenv(name); return e != nullptr ? std::atoi(e) : dflt; }

#define PCO_HIP_CHECK(expr)                                                                   \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) throw pcogfx::HostError{PCO_GFX_DEVICE_ERROR, std::string(#expr) + ": " + hipGetErrorString(_e)}; \
  } while (0)

// A growable device buffer (never shrinks; freed by pco_gfx_release_workspace, and when a thread other than the main one exits).
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  void* ensure(size_t bytes) {
    if (bytes > cap) {
      if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
      size_t want = bytes + std::min<size_t>(bytes / 8, (size_t)64 << 20) + 256;   // (slack against calls that grow a little; bounded: an eighth of 10 GB is real memory)
      hipError_t e = hipMalloc(&p, want);
      if (e != hipSuccess) { p = nullptr; (void)hipGetLastError(); throw HostError{PCO_GFX_DEVICE_ERROR, std::string("hipMalloc failed: ") + hipGetErrorString(e), true}; }
      cap = want;
    }
    return p;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// A growable pinned host buffer for read-backs (a fresh std::vector of 100 MB costs more in page faults than the copy).
struct HostBuf {
  void* p = nullptr;
  size_t cap = 0;
  void* ensure(size_t bytes) {
    if (bytes > cap) {
      if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
      size_t want = bytes + bytes / 8 + 256;
      hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
      if (e != hipSuccess) { p = nullptr; (void)hipGetLastError(); throw HostError{PCO_GFX_DEVICE_ERROR, std::string("hipHostMalloc failed: ") + hipGetErrorString(e), true}; }
      cap = want;
    }
    return p;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

struct Workspace {
  int device = -1;
  // One workspace per (thread, device), shared by every call whatever its stream: a call on stream B while an asynchronous call on
  // stream A still runs would overwrite buffers in use, so each call first makes its stream wait for the previous call's last
  // kernel (an event recorded at that call's end; free when both calls use one stream, the documented pattern).
  hipStream_t last_stream = nullptr; hipEvent_t last_event = nullptr; bool has_last = false;
  DevBuf tasks, results, tbl_ws;          // decode
  DevBuf dec_plans, dec_bins, dec_sym, dec_offpos;  // decode fast path
  DevBuf dec_progress;                               // ... and the walker -> trailing-expander hand-over words (decode_trail.hip)
  DevBuf dec_stats;                                  // device counter: chunks the trailing expanders gave back to dec_expand_kernel (pco_gfx_trail_givebacks)
  hipStream_t side_stream = nullptr, side_stream2 = nullptr; hipEvent_t fork_event = nullptr, join_event = nullptr, join_event2 = nullptr;   // the expanders' stream and the ordinary walker's (created on first use)
  int n_cus = 0;
  DevBuf dec_hist;                                   // decode: scratch for a delta'd secondary variable's lookback history (rare)
  DevBuf io_in, io_out;                   // staging for the host-buffer entry points
  DevBuf compact_tasks;                   // pco_gfx_compact_chunks
  DevBuf enc_state;                       // encode: per chunk plans etc. (see encode_kernels.hip)
  DevBuf enc_lat, enc_lat2, enc_sort, enc_ans, enc_small, enc_lb, enc_walk;
  DevBuf enc_lbprops;                     // lookback: the hash proposals of every page (six u16 streams), enc_lookback_hash_kernel -> enc_lookback_pipe_kernel
  DevBuf enc_sym, enc_answ, enc_bat, enc_run, enc_fstate, enc_vlut;   // encode fast path (encode_fast.hip)
  DevBuf enc_strict;                      // strict histograms: how many variables took the reference's heapsort branch (a device counter)
  DevBuf auto_idx, auto_samp, auto_tasks, auto_sum, auto_log2;  // Auto spec resolution
  HostBuf h_samp, h_sum;                             // ... and its read-backs
#define PCO_WS_BUFS(X) X(tasks) X(results) X(tbl_ws) X(dec_plans) X(dec_bins) X(dec_sym) X(dec_offpos) X(dec_progress) X(dec_stats) X(dec_hist) X(io_in) X(io_out) X(compact_tasks) X(enc_state) \
  X(enc_lat) X(enc_lat2) X(enc_sort) X(enc_ans) X(enc_small) X(enc_lb) X(enc_lbprops) X(enc_walk) X(enc_sym) X(enc_answ) X(enc_bat) X(enc_run) X(enc_fstate) X(enc_vlut) X(enc_strict) X(auto_idx) X(auto_samp) X(auto_tasks) X(auto_sum) X(auto_log2)
  size_t device_bytes() const {   // what the workspace holds on the device right now (pco_gfx_workspace_bytes)
    size_t b = 0;
#define PCO_WS_ADD(name) b += name.cap;
    PCO_WS_BUFS(PCO_WS_ADD)
#undef PCO_WS_ADD
    return b;
  }
  std::string device_report() const {   // "name=bytes ..." of the buffers that hold anything (PCO_GFX_TRACE)
    std::string r;
#define PCO_WS_REP(name) if (name.cap) r += std::string(#name) + "=" + std::to_string(name.cap) + " ";
    PCO_WS_BUFS(PCO_WS_REP)
#undef PCO_WS_REP
    return r;
  }
  // The two device counters (pco_gfx_trail_givebacks / _marked, pco_gfx_strict_histogram_fallbacks) count since the workspace was CREATED: what
  // they held is folded into 64-bit host totals before their buffers go (release_all also runs on the out-of-memory retry path of a call).
  unsigned long long acc_givebacks = 0, acc_marked = 0, acc_strict = 0;
  void fold_counters() {
    uint32_t v[2] = {0, 0};
    if (dec_stats.p && hipMemcpy(v, dec_stats.p, 8, hipMemcpyDeviceToHost) == hipSuccess) { acc_givebacks += v[0]; acc_marked += v[1]; }
    uint32_t f = 0;
    if (enc_strict.p && hipMemcpy(&f, enc_strict.p, 4, hipMemcpyDeviceToHost) == hipSuccess) acc_strict += f;
  }
  void release_all() {
    fold_counters();
    // (the side streams and their events belong to the workspace too: a worker thread that decoded >= 1024 chunks once must not leak them)
    if (side_stream) { (void)hipStreamSynchronize(side_stream); (void)hipStreamDestroy(side_stream); side_stream = nullptr; }
    if (side_stream2) { (void)hipStreamSynchronize(side_stream2); (void)hipStreamDestroy(side_stream2); side_stream2 = nullptr; }
    if (fork_event) { (void)hipEventDestroy(fork_event); fork_event = nullptr; }
    if (join_event) { (void)hipEventDestroy(join_event); join_event = nullptr; }
    if (join_event2) { (void)hipEventDestroy(join_event2); join_event2 = nullptr; }
    tasks.release(); results.release(); tbl_ws.release(); dec_plans.release(); dec_bins.release(); dec_sym.release(); dec_offpos.release(); dec_progress.release(); dec_stats.release(); dec_hist.release(); io_in.release(); io_out.release(); compact_tasks.release();
    enc_state.release(); enc_lat.release(); enc_lat2.release(); enc_sort.release(); enc_ans.release(); enc_small.release(); enc_lb.release(); enc_lbprops.release(); enc_walk.release(); enc_sym.release(); enc_answ.release(); enc_bat.release(); enc_run.release(); enc_fstate.release(); enc_vlut.release(); enc_strict.release(); auto_idx.release(); auto_samp.release(); auto_tasks.release(); auto_sum.release(); auto_log2.release(); h_samp.release(); h_sum.release();
  }
};
Workspace& workspace();
struct WorkspaceUse {   // RAII: orders the call after the workspace's previous user, records the hand-over event at scope exit
  Workspace& ws; hipStream_t stream;
  WorkspaceUse(Workspace& w, hipStream_t s) : ws(w), stream(s) {
    if (ws.has_last && ws.last_stream != stream && ws.last_event) (void)hipStreamWaitEvent(stream, ws.last_event, 0);
  }
  ~WorkspaceUse() {
    if (!ws.last_event && hipEventCreateWithFlags(&ws.last_event, hipEventDisableTiming) != hipSuccess) { ws.last_event = nullptr; return; }
    if (hipEventRecord(ws.last_event, stream) == hipSuccess) { ws.last_stream = stream; ws.has_last = true; }
  }
};

// ---- host-side bit writer for framing / size computations (O(metadata) work only) ----
struct HostBitWriter {
  std::vector<uint8_t> buf;
  uint64_t bit = 0;
  void write(uint64_t v, uint32_t n) {
    if (n == 0) return;
    if (n < 64) v &= ((uint64_t)1 << n) - 1;
    size_t need = (size_t)((bit + n + 7) >> 3) + 9;
    if (buf.size() < need) buf.resize(need * 2, 0);
    for (uint32_t i = 0; i < n;) {
      size_t byte = (size_t)(bit >> 3); uint32_t sh = (uint32_t)(bit & 7);
      uint32_t take = 8 - sh; if (take > n - i) take = n - i;
      buf[byte] |= (uint8_t)(((v >> i) & ((1u << take) - 1)) << sh);
      i += take; bit += take;
    }
  }
  void finish_byte() { bit = (bit + 7) & ~(uint64_t)7; }
  size_t bytes() const { return (size_t)((bit + 7) >> 3); }
};

// size guarantees (wrapped/guarantee.rs:11-37, standalone/guarantee.rs:11-37, metadata/chunk.rs:105-113)
size_t guarantee_wrapped_chunk_size(int latent_bits, size_t n);
size_t guarantee_standalone_chunk_size(int latent_bits, size_t n);
size_t guarantee_standalone_header_size();
bool n_per_page(uint64_t max_page_n, size_t n, std::vector<size_t>& out);  // chunk_config.rs:134-182

}  // namespace pcogfx
