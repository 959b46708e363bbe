// handoff_test.hip -- a producer kernel on one stream hands data to a consumer kernel on another stream WHILE BOTH RUN (the
// decode walker -> trailing expander hand-over): is the data the consumer reads fresh, on the same XCD and across XCDs?
//   producer: 1024 blocks x 64 lanes, 40928 B of LDS (four per CU like dec_walk_kernel).  Round r: every lane stores 16 bytes
//             f(block, r, lane) to data[block][r][lane], waits for its stores (s_waitcnt vmcnt(0)), lane 0 publishes progress[block] = r + 1.
//   consumer: 1024 blocks x 256 lanes, no LDS.  Wave w of block c follows producer block (c + shift) % 1024: polls progress (bounded),
//             reads round r's 1024 bytes, compares with f; counts mismatches (= stale or torn data) and time-outs.
//   mode 0: data and progress through agent-scope relaxed atomics (sc1 stores / loads); mode 1: data through plain stores / loads.
// Also prints which XCC the first blocks of each kernel ran on (s_getreg HW_REG_XCC_ID).
// build: hipcc --offload-arch=gfx950 -O3 scripts/micro/handoff_test.hip -o scripts/micro/handoff_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint64_t f(uint32_t b, uint32_t r, uint32_t lane, uint32_t half) { return ((uint64_t)(b * 2654435761u + r * 40503u + lane * 97u + half) << 20) ^ (uint64_t)r ^ ((uint64_t)half << 63); }
__device__ __forceinline__ uint32_t xcc_id() { uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }

template <int MODE>
__global__ __launch_bounds__(64) void producer(uint64_t* data, uint32_t* progress, uint32_t rounds, uint32_t spin, uint32_t* xcc_out) {
  extern __shared__ uint32_t lds[];
  const uint32_t lane = threadIdx.x, b = blockIdx.x;
  lds[lane] = lane;
  if (lane == 0) xcc_out[b] = xcc_id();
  for (uint32_t r = 0; r < rounds; r++) {
    uint32_t p = lane; for (uint32_t s = 0; s < spin; s++) p = lds[p & 63] + 1;   // the latency chain between two batches
    uint64_t* o = data + ((size_t)b * rounds + r) * 128 + 2 * lane;
    const uint64_t v0 = f(b, r, lane, 0) + (p & 0), v1 = f(b, r, lane, 1);
    if (MODE == 0) { __hip_atomic_store(o, v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(o + 1, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    else { o[0] = v0; o[1] = v1; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(progress + b, r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void consumer(const uint64_t* data, const uint32_t* progress, uint32_t rounds, uint32_t shift, uint32_t* bad, uint32_t* xcc_out) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t b = (blockIdx.x + shift) % gridDim.x;
  if (threadIdx.x == 0) xcc_out[blockIdx.x] = xcc_id();
  uint32_t nbad = 0, ntimeout = 0;
  for (uint32_t r = wave; r < rounds; r += 4) {   // the block's four waves share the producer's rounds
    uint32_t tries = 0;
    while (__hip_atomic_load(progress + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= r) { __builtin_amdgcn_s_sleep(16); if (++tries > (1u << 22)) { ntimeout = 1; break; } }
    if (ntimeout) break;
    const uint64_t* o = data + ((size_t)b * rounds + r) * 128 + 2 * lane;
    uint64_t v0, v1;
    if (MODE == 0) { v0 = __hip_atomic_load(o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v1 = __hip_atomic_load(o + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    else { v0 = o[0]; v1 = o[1]; }
    if (v0 != f(b, r, lane, 0) || v1 != f(b, r, lane, 1)) nbad++;
  }
  if (nbad) atomicAdd(bad, nbad);
  if (ntimeout && lane == 0) atomicAdd(bad + 1, 1u);
}

template <int MODE> int run(uint32_t rounds, uint32_t spin, uint32_t shift) {
  const uint32_t nb = 1024;
  uint64_t* data; uint32_t *progress, *bad, *xp, *xc;
  CK(hipMalloc(&data, (size_t)nb * rounds * 1024)); CK(hipMalloc(&progress, nb * 4)); CK(hipMalloc(&bad, 8)); CK(hipMalloc(&xp, nb * 4)); CK(hipMalloc(&xc, nb * 4));
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e0, e1, j; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&j, hipEventDisableTiming));
  CK(hipFuncSetAttribute((const void*)producer<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 40928));
  for (int rep = 0; rep < 3; rep++) {
    CK(hipMemset(data, 0xff, (size_t)nb * rounds * 1024)); CK(hipMemset(progress, 0, nb * 4)); CK(hipMemset(bad, 0, 8));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, s1));
    hipLaunchKernelGGL(producer<MODE>, dim3(nb), dim3(64), 40928, s1, data, progress, rounds, spin, xp);
    CK(hipStreamWaitEvent(s2, e0, 0));
    hipLaunchKernelGGL(consumer<MODE>, dim3(nb), dim3(256), 0, s2, data, progress, rounds, shift, bad, xc);
    CK(hipEventRecord(j, s2)); CK(hipStreamWaitEvent(s1, j, 0));
    CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    uint32_t hb[2], hx[16], hy[16]; CK(hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hx, xp, 64, hipMemcpyDeviceToHost)); CK(hipMemcpy(hy, xc, 64, hipMemcpyDeviceToHost));
    printf("mode %d shift %u rounds %u: %.3f ms, mismatching lanes %u, timed-out waves %u | producer xcc", MODE, shift, rounds, ms, hb[0], hb[1]);
    for (int i = 0; i < 10; i++) printf(" %u", hx[i]); printf(" | consumer xcc"); for (int i = 0; i < 10; i++) printf(" %u", hy[i]); printf("\n");
  }
  return 0;
}

int main(int argc, char** argv) {
  const uint32_t rounds = argc > 1 ? atoi(argv[1]) : 1024, spin = argc > 2 ? atoi(argv[2]) : 60;
  for (uint32_t shift : {0u, 3u}) { if (run<0>(rounds, spin, shift)) return 1; if (run<1>(rounds, spin, shift)) return 1; }
  return 0;
}
