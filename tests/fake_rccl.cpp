// fake_rccl.cpp -- TEST INFRASTRUCTURE, not part of the product: a loopback stand-in for the nine RCCL entry points libpco_gfx.so
// dlopens (pcodec_amd/csrc/pco_gfx_comm.inc), selected with PCO_GFX_RCCL_LIB=tests/fake_rccl.so.
//
// Why it exists: RCCL refuses two ranks on one device ("Duplicate GPU detected") and the round's GPU box has one MI355X, so the
// rank != root branches of pco_gfx_gather_chunks / pco_gfx_scatter_chunks (offset arithmetic, group posting, zero-byte ranks, a
// non-zero root, the collective failure path) would first execute on an 8-GPU node.  With this transport N PROCESSES SHARING ONE
// DEVICE run exactly the product code above the nccl* calls; only the bytes travel differently (device -> host file -> device).
//
// Transport: a directory named by the unique id (under $PCO_FAKE_RCCL_DIR or /dev/shm).  A message is a file written under a
// temporary name and renamed into place, so a reader that sees it sees all of it:
//   ag_<seq>_<rank>          one rank's contribution to the seq-th all-gather (every rank reads all n of them)
//   p2p_<src>_<dst>_<seq>    the seq-th message from src to dst (the receiver deletes it)
// ncclSend never blocks (it writes the file), so a group of sends and receives cannot deadlock whatever the posting order; every call
// synchronises the stream it was given first (the real calls are stream-ordered; here the copies are synchronous).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <sys/stat.h>
#include <unistd.h>

typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
typedef int ncclDataType_t;
struct ncclComm {
  std::string dir;
  int n = 0, rank = 0;
  uint64_t ag_seq = 0;
  std::vector<uint64_t> send_seq, recv_seq;   // per peer
};
typedef ncclComm* ncclComm_t;

namespace {
constexpr ncclResult_t kOk = 0, kSystemError = 2, kInvalidArgument = 4;
constexpr double kTimeoutS = 120.0;

size_t type_bytes(ncclDataType_t t) {
  switch (t) {
    case 0: case 1: return 1;      // ncclInt8 / ncclUint8
    case 2: case 3: case 7: return 4;   // ncclInt32 / ncclUint32 / ncclFloat32
    case 4: case 5: case 8: return 8;   // ncclInt64 / ncclUint64 / ncclFloat64
    case 6: case 9: return 2;      // ncclFloat16 / ncclBfloat16
    default: return 0;
  }
}
bool write_file(const std::string& path, const void* p, size_t len) {
  const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) return false;
  const bool ok = len == 0 || fwrite(p, 1, len, f) == len;
  fclose(f);
  return ok && rename(tmp.c_str(), path.c_str()) == 0;
}
bool read_file_when_there(const std::string& path, void* p, size_t len) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    struct stat st;
    if (stat(path.c_str(), &st) == 0) {
      if ((size_t)st.st_size != len) return false;   // the partner sent another size than this rank expects: a bug in the caller
      FILE* f = fopen(path.c_str(), "rb");
      if (!f) return false;
      const bool ok = len == 0 || fread(p, 1, len, f) == len;
      fclose(f);
      return ok;
    }
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kTimeoutS) return false;
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
}
std::string base_dir() {
  const char* e = std::getenv("PCO_FAKE_RCCL_DIR");
  return e && *e ? std::string(e) : std::string("/dev/shm");
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  std::memset(id, 0, sizeof(*id));
  const unsigned long long t = (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count();
  snprintf(id->internal, sizeof(id->internal), "pco_fake_rccl_%ld_%llx", (long)getpid(), t);
  return kOk;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int n, ncclUniqueId id, int rank) {
  if (!out || n < 1 || rank < 0 || rank >= n) return kInvalidArgument;
  id.internal[sizeof(id.internal) - 1] = 0;
  ncclComm* c = new ncclComm();
  c->dir = base_dir() + "/" + id.internal; c->n = n; c->rank = rank;
  c->send_seq.assign(n, 0); c->recv_seq.assign(n, 0);
  mkdir(c->dir.c_str(), 0700);   // (whoever comes first; EEXIST is fine)
  // rendezvous, like the real call: nobody returns before everybody is here
  char one = 1;
  if (!write_file(c->dir + "/hello_" + std::to_string(rank), &one, 1)) { delete c; return kSystemError; }
  for (int r = 0; r < n; r++) if (!read_file_when_there(c->dir + "/hello_" + std::to_string(r), &one, 1)) { delete c; return kSystemError; }
  *out = c;
  return kOk;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return kOk;
  // leave the directory to the last rank out: each rank removes its own leftovers, rmdir succeeds for whoever empties it
  unlink((c->dir + "/hello_" + std::to_string(c->rank)).c_str());
  for (uint64_t s = c->ag_seq >= 2 ? c->ag_seq - 2 : 0; s < c->ag_seq; s++) unlink((c->dir + "/ag_" + std::to_string(s) + "_" + std::to_string(c->rank)).c_str());
  rmdir(c->dir.c_str());
  delete c;
  return kOk;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t type, ncclComm_t c, hipStream_t stream) {
  const size_t len = count * type_bytes(type);
  if (!c || !len) return kInvalidArgument;
  if (hipStreamSynchronize(stream) != hipSuccess) return kSystemError;
  std::vector<char> mine(len), all(len * c->n);
  if (hipMemcpy(mine.data(), send, len, hipMemcpyDeviceToHost) != hipSuccess) return kSystemError;
  const uint64_t s = c->ag_seq++;
  if (!write_file(c->dir + "/ag_" + std::to_string(s) + "_" + std::to_string(c->rank), mine.data(), len)) return kSystemError;
  for (int r = 0; r < c->n; r++)
    if (!read_file_when_there(c->dir + "/ag_" + std::to_string(s) + "_" + std::to_string(r), all.data() + len * r, len)) return kSystemError;
  // every rank has read round s - 2 by the time anybody writes round s (it had to finish s - 1, which needed everyone's s - 1,
  // which they wrote after reading s - 2): my file of that round can go
  if (s >= 2) unlink((c->dir + "/ag_" + std::to_string(s - 2) + "_" + std::to_string(c->rank)).c_str());
  if (hipMemcpy(recv, all.data(), all.size(), hipMemcpyHostToDevice) != hipSuccess) return kSystemError;
  return kOk;
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t c, hipStream_t stream) {
  const size_t len = count * type_bytes(type);
  if (!c || peer < 0 || peer >= c->n || peer == c->rank) return kInvalidArgument;
  if (hipStreamSynchronize(stream) != hipSuccess) return kSystemError;
  std::vector<char> h(len);
  if (len && hipMemcpy(h.data(), buf, len, hipMemcpyDeviceToHost) != hipSuccess) return kSystemError;
  const uint64_t s = c->send_seq[peer]++;
  return write_file(c->dir + "/p2p_" + std::to_string(c->rank) + "_" + std::to_string(peer) + "_" + std::to_string(s), h.data(), len) ? kOk : kSystemError;
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t c, hipStream_t stream) {
  const size_t len = count * type_bytes(type);
  if (!c || peer < 0 || peer >= c->n || peer == c->rank) return kInvalidArgument;
  if (hipStreamSynchronize(stream) != hipSuccess) return kSystemError;
  std::vector<char> h(len);
  const uint64_t s = c->recv_seq[peer]++;
  const std::string path = c->dir + "/p2p_" + std::to_string(peer) + "_" + std::to_string(c->rank) + "_" + std::to_string(s);
  if (!read_file_when_there(path, h.data(), len)) return kSystemError;
  unlink(path.c_str());
  if (len && hipMemcpy(buf, h.data(), len, hipMemcpyHostToDevice) != hipSuccess) return kSystemError;
  return kOk;
}

ncclResult_t ncclGroupStart() { return kOk; }
ncclResult_t ncclGroupEnd() { return kOk; }
const char* ncclGetErrorString(ncclResult_t r) { return r == kOk ? "ok" : (r == kInvalidArgument ? "fake rccl: invalid argument" : "fake rccl: transport error (timeout, size mismatch or HIP failure)"); }

}  // extern "C"
