// overlap_test.hip -- do two kernels from two streams share the CUs?  (the question behind running the decode expanders UNDER the tANS
// walker: the walker is a latency chain that fills the LDS and leaves the SIMDs nine tenths idle.)
//   A: 1024 blocks of one wave, 40928 B of LDS each (four per CU, like dec_walk_kernel), a dependent LDS pointer chase of `steps` steps
//   B: an LDS-free streaming kernel (reads 4 B, writes 16 B per thread-iteration, like dec_expand's traffic), grid-stride, <= 64 VGPRs
// Prints A alone, B alone, A then B on one stream, A || B on two streams.
// build: hipcc --offload-arch=gfx950 -O3 scripts/micro/overlap_test.hip -o scripts/micro/overlap_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(64) void chase_kernel(uint32_t* out, uint32_t steps) {
  extern __shared__ uint32_t lds[];
  const uint32_t n = 40928 / 4;
  for (uint32_t i = threadIdx.x; i < n; i += 64) lds[i] = (i * 2654435761u + 12345u) % n;
  __syncthreads();
  uint32_t p = threadIdx.x;
  for (uint32_t s = 0; s < steps; s++) { p = lds[p]; p = (p * 3u + (p >> 3)) % n; }   // LDS round trip + a short dependent VALU stretch
  out[blockIdx.x * 64 + threadIdx.x] = p;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void stream_kernel(const uint32_t* __restrict__ in, u32x4* __restrict__ outp, size_t n, uint32_t work) {
  const size_t stride = (size_t)gridDim.x * 64;
  for (size_t i = (size_t)blockIdx.x * 64 + threadIdx.x; i < n; i += stride) {
    uint32_t v = in[i];
    uint32_t a = v, b = v ^ 0x9e3779b9u;
    for (uint32_t w = 0; w < work; w++) { a = a * 1664525u + b; b ^= a >> 7; }   // `work` x 3 dependent VALU per element-quad
    u32x4 o; o.x = a; o.y = b; o.z = a + b; o.w = a ^ b;
    outp[i] = o;
  }
}

int main(int argc, char** argv) {
  const uint32_t steps = argc > 1 ? atoi(argv[1]) : 40000;
  const uint32_t work = argc > 2 ? atoi(argv[2]) : 16;
  const int bgrid = argc > 3 ? atoi(argv[3]) : 7168;
  const size_t n = (size_t)1 << 30;   // 4 GiB read, 16 GiB written
  uint32_t *in, *out; u32x4* big;
  CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&big, n * 16)); CK(hipMalloc(&out, 1024 * 64 * 4));
  CK(hipMemset(in, 1, n * 4));
  CK(hipFuncSetAttribute((const void*)chase_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 40928));
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* what, int mode) -> int {
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, s1));
      if (mode == 0 || mode == 2 || mode == 3) hipLaunchKernelGGL(chase_kernel, dim3(1024), dim3(64), 40928, s1, out, steps);
      if (mode == 1 || mode == 2) hipLaunchKernelGGL(stream_kernel, dim3(bgrid), dim3(64), 0, s1, in, big, n, work);
      if (mode == 3) {   // B on the second stream, started after e0, joined before e1
        CK(hipStreamWaitEvent(s2, e0, 0));
        hipLaunchKernelGGL(stream_kernel, dim3(bgrid), dim3(64), 0, s2, in, big, n, work);
        hipEvent_t j; CK(hipEventCreateWithFlags(&j, hipEventDisableTiming)); CK(hipEventRecord(j, s2)); CK(hipStreamWaitEvent(s1, j, 0)); CK(hipEventDestroy(j));
      }
      CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%-34s %8.3f ms\n", what, best);
    return 0;
  };
  printf("chase steps %u, stream work %u, stream grid %d x 64\n", steps, work, bgrid);
  if (timeit("A alone (LDS chase, 4 blocks/CU)", 0)) return 1;
  if (timeit("B alone (stream 4+16 GiB)", 1)) return 1;
  if (timeit("A then B, one stream", 2)) return 1;
  if (timeit("A || B, two streams", 3)) return 1;
  return 0;
}
